// lsa_slam_core.h -- per-frame pipeline of LidarSlam::Slam on top of the C ABI
// (slam_lib/src/Slam.cxx: AddFrames :230-344, ExtractKeypoints :746-810, ComputeEgoMotion
//  :813-972, Localization :975-1175, UpdateMapsUsingTworld :1178-1222, LogCurrentFrameState
//  :1225-1264, undistortion :1271-1352; defaults slam_lib/include/LidarSlam/Slam.h:403-694).
//
// Host keeps what SURVEY.md 8 leaves on the host: frame checks, pose bookkeeping, the
// 6-dof trust-region control flow, the rolling voxel maps.  Everything per point / per
// keypoint runs on the device through lidarslam_amd.h.
#pragma once
#include <array>
#include <atomic>
#include <condition_variable>
#include <deque>
#include <limits>
#include <map>
#include <functional>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>
#include "../../../include/lidarslam_amd.h"
#include "lsa_hostmath.h"
#include "lsa_lm.h"
#include "lsa_rolling_grid.h"

namespace lsa
{
namespace host
{

enum UndistortionMode { UNDISTORTION_NONE = 0, UNDISTORTION_ONCE = 1, UNDISTORTION_REFINED = 2 };
enum class EgoMotionMode { NONE = 0, MOTION_EXTRAPOLATION = 1, REGISTRATION = 2, MOTION_EXTRAPOLATION_AND_REGISTRATION = 3 };
enum class MappingMode { NONE = 0, ADD_KPTS_TO_FIXED_MAP = 1, UPDATE = 2 };

struct StampedPose
{
  Pose pose;
  double time;
};

struct FrameStats
{
  double total = 0, extract = 0, ego_icp = 0, ego_lm = 0, loc_icp = 0, loc_lm = 0, undistort = 0, submap = 0, maps = 0;
  int ego_iters = 0, loc_iters = 0, lm_evals = 0;
  int submap_spec_hits = 0;  // sub-maps extracted ahead of time that Localization() could keep
  double maps_wait = 0;   // time this frame waited for the previous keyframe's map insertion
  double maps_async = 0;  // duration of that insertion on the worker thread
};

struct MatchDebug
{
  std::vector<uint8_t> status;
  std::vector<double> weights;
};

// One persistent host thread that runs jobs in submission order.  Each map has one: the keyframe's
// insertion (RollingGrid::Add, hash-map bound) is handed to it at the end of AddFrame and overlaps the next
// frame's keypoint extraction and ego-motion ICP on the device, and the sub-maps of the three keypoint
// types are extracted side by side.  Wait() is called before anything else reads a map.
// Same operations in the same order on the same containers: results do not depend on the overlap.
class HostWorker
{
public:
  HostWorker();
  ~HostWorker();
  void Submit(std::function<void()> job);
  void Wait();
  // A job is about to be submitted (within `seconds`): the thread wakes up now and polls for it instead of sleeping, so that
  // the job starts within a microsecond of Submit instead of a scheduler's wake-up later (tens of microseconds on an idle
  // host, a fifth of a frame and more on a busy one: a map insertion enqueued late runs INTO the next frame's ICP kernels
  // instead of in front of them).  Costs the polling time on one core, bounded by `seconds`.
  void Expect(double seconds);

private:
  void Run();
  std::mutex M;
  std::condition_variable Cv, Idle;
  std::deque<std::function<void()>> Jobs;
  bool Busy = false, Quit = false;
  std::atomic<int> Posted{0};           // jobs in the queue (read without the lock by the polling thread)
  std::atomic<long long> PollUntil{0};  // steady-clock nanoseconds
  std::atomic<bool> Leaving{false};
  std::thread T;
};

class SlamCore
{
public:
  explicit SlamCore(int device);
  ~SlamCore();
  SlamCore(const SlamCore&) = delete;
  SlamCore& operator=(const SlamCore&) = delete;

  bool Ok() const { return Ctx != nullptr; }
  const std::string& Error() const { return LastError; }
  lsa_ctx* Context() { return Ctx; }

  void Reset(bool resetLog = true);
  // Slam::AddFrame on a host scan / on a scan resident in the frame store
  int AddFrame(const lsa_point_t* pts, int n, uint64_t stampUs, uint32_t seq);
  int AddStoredFrame(int slot, uint64_t stampUs, uint32_t seq);
  // Replay: the frame-store slot of the frame that will be added next.  Its keypoints are then extracted beside
  // the registration of the current frame (lsa_extract_prefetch) and the next AddStoredFrame finds them ready.
  // Slam::ClearMaps (Slam.cxx:1639-1643)
  void ClearMaps()
  {
    WaitMaps();
    for (auto& m : LocalMaps) m->Reset();
    for (auto* g : DevMaps)
      if (g) lsa_device_grid_reset(g, nullptr);
  }
  void HintNextStoredFrame(int slot) { NextStoredSlot = slot; }
  // Replay from host clouds: the cloud of the AddFrame call after the next one.  Its upload starts at once (pinned
  // staging, copy stream, a thread of its own), its keypoints are extracted beside the registration of the frame in
  // between; AddFrame adopts both when it is handed this very cloud.  The cloud must stay valid until then.
  int HintNextFrame(const lsa_point_t* pts, int n);
  // Slam::AddFrames with one frame per LiDAR device (Slam.cxx:230-344, 753-801)
  struct InputFrame
  {
    const lsa_point_t* pts;
    int n;
    uint64_t stampUs;
    uint32_t seq;
  };
  int AddFrames(const InputFrame* frames, int nframes);
  // Slam::SetKeyPointsExtractor(extractor, deviceId) / SetBaseToLidarOffset(transform, deviceId) (Slam.h:239-250):
  // device 0 always has an extractor (ExtractParams); the others get one with their first parameter
  int SetExtractorParam(int deviceId, const std::string& name, double v);
  int GetExtractorParam(int deviceId, const std::string& name, double* v) const;
  int SetBaseToLidarOffset(int deviceId, const Pose& offset);
  Pose GetBaseToLidarOffset(int deviceId) const;

  Pose GetWorldTransform(double* time = nullptr) const;
  // Slam::GetLatencyCompensatedWorldTransform (Slam.cxx:555-590): the last pose extrapolated by Latency
  Pose GetLatencyCompensatedWorldTransform(double* time = nullptr) const;
  // Slam::SetWorldTransformFromGuess (Slam.cxx:490-501)
  int SetWorldTransformFromGuess(const Pose& guess);
  // Slam::GetDebugInformation (Slam.cxx:610-633): matches used by the last ego-motion / localization iteration per
  // type, registration errors, overlap, motion-limit compliance.  Reads five 32-byte histograms back on demand.
  int GetDebugInformation(double out[10]);
  const std::array<double, 36>& GetTransformCovariance() const { return LocalizationUncertainty.Covariance; }
  int GetKeypoints(int type, bool world, std::vector<lsa_point_t>& out);
  int GetRawKeypoints(int type, std::vector<lsa_point_t>& out);
  int GetRegisteredFrame(std::vector<lsa_point_t>& out);
  const MatchDebug& GetMatchDebug(bool localization, int type) const { return localization ? LocDebug[type] : EgoDebug[type]; }

  int SetParam(const std::string& name, double v);
  int GetParam(const std::string& name, double* v) const;
  // LocalMaps[k] after every pending insertion has landed (Slam::GetMap reads through this)
  RollingGrid& Map(int k) { WaitMaps(); return *LocalMaps[k]; }

  // ---- parameters (names = the reference's members) ----
  bool UseKeypoints[3] = {true, true, false};
  EgoMotionMode EgoMotion = EgoMotionMode::MOTION_EXTRAPOLATION;
  UndistortionMode Undistortion = UNDISTORTION_REFINED;
  bool TwoDMode = false;
  unsigned EgoMotionICPMaxIter = 4, LocalizationICPMaxIter = 3;
  unsigned EgoMotionLMMaxIter = 15, LocalizationLMMaxIter = 15;
  double EgoMotionMaxNeighborsDistance = 5., LocalizationMaxNeighborsDistance = 5.;
  unsigned EgoMotionEdgeNbNeighbors = 8, EgoMotionEdgeMinNbNeighbors = 3;
  double EgoMotionEdgeMaxModelError = 0.2;
  unsigned LocalizationEdgeNbNeighbors = 10, LocalizationEdgeMinNbNeighbors = 4;
  double LocalizationEdgeMaxModelError = 0.2;
  unsigned EgoMotionPlaneNbNeighbors = 5;
  double EgoMotionPlanarityThreshold = 0.04, EgoMotionPlaneMaxModelError = 0.2;
  unsigned LocalizationPlaneNbNeighbors = 5;
  double LocalizationPlanarityThreshold = 0.04, LocalizationPlaneMaxModelError = 0.2;
  unsigned LocalizationBlobNbNeighbors = 10;
  double EgoMotionInitSaturationDistance = 5., EgoMotionFinalSaturationDistance = 1.;
  double LocalizationInitSaturationDistance = 2., LocalizationFinalSaturationDistance = 0.5;
  double MaxExtrapolationRatio = 3.;
  unsigned MinNbMatchedKeypoints = 20;
  double KfDistanceThreshold = 0.5, KfAngleThreshold = 5.;
  MappingMode MapUpdate = MappingMode::UPDATE;
  // Confidence estimator: share of the frame's points with a map point nearby (Slam::SetOverlapSamplingRatio,
  // GetOverlapEstimation; 0 = off, the library default; the ROS configuration uses 0.33)
  float OverlapSamplingRatio = 0.f;
  float OverlapEstimation = -1.f;
  // Confidence estimator: velocity / acceleration over a time window against limits (Slam::CheckMotionLimits,
  // Slam.cxx:1391-1484; linear [m/s, m/s2] and angular [deg/s, deg/s2]); TimeWindowDuration = 0 turns it off
  float VelocityLimits[2] = {std::numeric_limits<float>::max(), std::numeric_limits<float>::max()};
  float AccelerationLimits[2] = {std::numeric_limits<float>::max(), std::numeric_limits<float>::max()};
  float TimeWindowDuration = 0.f;
  bool ComplyMotionLimits = true;
  // Slam::LoggingTimeout (Slam.h:425-438): 0 keeps the last two poses, > 0 the poses of that many seconds, < 0 all
  double LoggingTimeout = 0.;
  double Latency = 0.;  // duration of the last AddFrame [s] (Slam::GetLatency)
  Pose BaseToLidarOffset = Pose::Identity();  // device 0
  lsa_extract_params_t ExtractParams;         // device 0
  struct DeviceExtractor
  {
    lsa_extract_params_t params;
    float azimuthalResolution = 0.f;  // estimated from that device's first frame
  };
  std::map<int, DeviceExtractor> OtherExtractors;  // devices != 0
  std::map<int, Pose> OtherBaseToLidarOffsets;
  // edge length of the finest kNN search-grid cells (an implementation knob: results do not depend on it)
  double KnnCellSizeEgoMotion = 0.25;      // [m] previous-scan plane targets (dense)
  double KnnCellSizeEgoMotionEdges = 0.5;  // [m] previous-scan edge targets (sparse)
  double KnnCellScaleMaps = 1.0;           // x map leaf size, plane / blob sub-maps
  double KnnCellScaleMapsEdges = 2.5;      // x map leaf size, edge sub-maps
  // lanes per query in the first kNN kernel (sparse edge targets: more lanes, fewer queries per wavefront)
  int KnnLanesEdges = 16, KnnLanesPlanes = 8, KnnLanesBlobs = 8;
  int KnnRoundsEdges = 2, KnnRoundsPlanes = 2, KnnRoundsBlobs = 2;  // rounds of the first kNN kernel (2 or 3)
  // build the next frame's ego-motion targets beside this frame's registration (a scheduling knob: same results)
  bool BuildTargetsAhead = true;
  // LocalOptimizer::Solve as one launch (the trust-region loop on the device) instead of one launch per evaluation
  bool DeviceLM = true;
  // one ICP iteration's matching step as one launch (search + model fit fused, all types) instead of staged kernels
  bool FusedMatch = true;
  bool KeepMatchDebug = false;  // download MatchingResults::Rejections/Weights every frame (Slam::GetDebugArray)

  std::shared_ptr<RollingGrid> LocalMaps[3];
  // The same three maps resident on the device (lsa_device_grid: SURVEY.md 8f-1).  "MapsOnDevice" (default): keyframes
  // are inserted and sub-maps extracted without leaving the device -- no staging of the keypoints in host memory, no map
  // threads, no sub-map upload.  Off: the host containers above.  Both
  // hand their points out in key order ("OrderedMaps"), so the two give the same sub-maps and the same poses.
  lsa_device_grid* DevMaps[3] = {nullptr, nullptr, nullptr};
  bool MapsOnDevice = true;
  // device maps: sub-maps extracted for the predicted boxes beside the ego-motion ICP ("SubMapsAhead"): the extraction and
  // the spare target's search grid queue on the look-ahead stream behind the previous keyframe's insertions, a host thread
  // of their own enqueues them, and the localization swaps the target in when the actual box touches the same outer voxels
  // (980 against 950 frames/s; with the first, slower insertion kernels it lost: the sub-map came 0.05 ms late)
  bool SubMapsAhead = true;
  bool SubMapsAheadAdaptive = true;  // give it up for a while when the localization had to wait for it twice in a row
  bool LocalizationStartFused = true;  // reset + first undistortion + keypoint boxes of the localization as one launch (device maps)
  int IcpGateTimeouts = 0;        // ICP iterations enqueued ahead whose gate gave up (diagnostics; 0 on a healthy run)
  int ICPAhead = 2;               // ICP iterations enqueued ahead of their inputs.  2: the whole loop at once, every solve leaving the next iteration's pose on the device (lsa_icp_link); 1: iteration i + 1 behind a gate the host answers (lsa_icp_gate); 0: off
  bool UndistortInSearch = true;  // RefineUndistortion between two localization iterations inside the next iteration's search kernel
  bool SpecBoxesOnLookahead = true;  // the predicted boxes of the sub-maps ahead of time on the look-ahead stream, not the context's
  bool SpecGridsTogether = true;  // the search grids of the sub-maps extracted ahead of time built by one sequence of launches
  bool DevSpec[3] = {false, false, false};
  bool OrderedMaps = true;
  bool DeviceMapsInUse() const { return MapsOnDevice && DevMaps[0]; }  // (every sampling mode: CENTROID since round 3)
  int MigrateMaps(bool fromDevice);
  int SetParamValue(const std::string& name, double v);
  int GetMap(int k, bool clean, std::vector<lsa_point_t>& out);
  int GetTargetSubMap(int k, std::vector<lsa_point_t>& out);

  // ---- state ----
  Pose Tworld, PreviousTworld, Trelative;
  RegistrationError LocalizationUncertainty;
  unsigned TotalMatchedKeypoints = 0;
  unsigned NbrFrameProcessed = 0;
  unsigned SubMapSpecHitsTotal = 0;
  int KfCounter = 0;
  FrameStats Stats;
  std::deque<StampedPose> LogTrajectory;
  std::deque<std::array<double, 36>> LogCovariances;
  int KeypointCounts[3] = {0, 0, 0};

private:
  int ProcessCurrentFrame(uint64_t stampUs);
  int ExtractKeypoints();
  int ComputeEgoMotion();
  int Localization();
  int UpdateMapsUsingTworld();
  int EstimateOverlap();
  void CheckMotionLimits();
  float PreviousVelocity[2] = {0.f, 0.f};
  long long EgoMatchSerial[3] = {0, 0, 0}, LocMatchSerial[3] = {0, 0, 0};  // lsa_match_serial after the last iteration
  void LogCurrentFrameState(double time);
  Pose InterpolateScanPose(double time) const;
  int InitUndistortion();
  void InitUndistortion(double t0, double t1);
  int RefineUndistortion(Pose* outD0 = nullptr, Pose* outD1 = nullptr);
  int Fail(int rc, const char* where);
  lsa_match_params_t EgoMatchParams() const;
  lsa_match_params_t LocMatchParams() const;

  // the frames of the current AddFrames call when there are several (device-resident, in the last slots of the
  // context's frame store); empty for the single-frame calls, whose frame is the context's current one
  struct HeldFrame
  {
    int slot;
    int n;
    int device;
    double timeOffset;  // stamp - first frame's stamp [s]
  };
  std::vector<HeldFrame> CurrentFrames;
  float Device0AzimuthalResolution = 0.f;  // parked here while another device's frame is extracted
  int ExtractFrames();
  int PrepareNextEgoMotionTargets();
  void ArmLookaheadInterlude();
  int InterludeWork();
  int DevSpecStatus = 0;   // written by the look-ahead thread, read once DevSpecRunning is false
  std::atomic<bool> DevSpecRunning{false};
  int DevSpecBackoff = 0;  // frames the sub-maps ahead of time are not tried for (they were late)
  int DevSpecLate = 0;     // frames in a row the localization had to wait for them
  int DevSpecGood = 0, DevSpecPenalty = 4;
  int FinishLookaheadInterlude();
  bool InterludeRan = true;
  int InterludeStatus = 0;
  int NextStoredSlot = -1;
  bool NextFrameHinted = false;   // a cloud was announced (HintNextFrame) and its look-ahead extraction not started yet
  double DbgAcc[8] = {0, 0, 0, 0, 0, 0, 0, 0};  // (diagnostics, LSA_STAGE_DEBUG=1: seconds outside the stage timers, printed when the object goes)
  long DbgFrames = 0;
  bool WorkerPrewake = true;      // the worker threads are woken ahead of the jobs whose time is known (HostWorker::Expect)
  bool HoldLookahead = false;     // ... and must not start yet: the sub-maps extracted ahead of time for THIS frame's localization go onto the look-ahead stream first
  int TryStartLookahead();        // starts it as soon as the upload has been enqueued
  lsa_ctx* Ctx = nullptr;
  std::string LastError;
  uint64_t CurrentStamp = 0;
  bool HaveFrame = false;
  double CurrentTime = 0.;
  WithinFrameMotion Motion;
  Pose KfLastPose;
  MatchDebug EgoDebug[3], LocDebug[3];
  std::vector<lsa_point_t> Scratch;
  // Sub-maps are extracted ahead of time: as soon as the keypoints exist, their bounding box under the
  // PREDICTED pose is reduced on the device (asynchronously) and the map workers extract the sub-maps for it
  // while the ego-motion ICP runs.  Localization() checks the box under the actual pose: the sub-map only
  // depends on the range of 10 m voxels the box touches, which a pose refinement rarely changes; if it did,
  // the sub-map is extracted again.  Same result either way.
  int BeginSubMapSpeculation(const Pose& predicted);
  int FinishSubMapSpeculation();
  bool SpecPending = false;
  bool SpecBuilt[3] = {false, false, false};  // written by the workers, read after WaitMaps
  // the same news for the thread that runs the ICP: it hands a finished sub-map to the device (upload and search
  // grid on the look-ahead stream) at its next stop between two kernels' results
  std::atomic<bool> SpecDone[3];
  bool SpecStaged[3] = {false, false, false};
  int StageSpeculativeSubMaps();
  double MapJobSeconds[3] = {0., 0., 0.};  // written by the workers, read after WaitMaps
  HostWorker AheadWorker;                   // enqueues the next frame's ego-motion targets beside the first solve
  int AheadStatus = 0;                  // result of the work done between a solve's launch and its wait
  int MapJobFailed[3] = {0, 0, 0};          // a device-map insertion that failed on its worker (read after WaitMaps)
  HostWorker MapWorker[3];                  // one per map: the three rolling grids are independent
  void WaitMaps() { for (auto& w : MapWorker) w.Wait(); AheadWorker.Wait(); }
};

}  // namespace host
}  // namespace lsa
