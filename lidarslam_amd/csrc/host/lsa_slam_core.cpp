// lsa_slam_core.cpp -- see lsa_slam_core.h.
#include "lsa_slam_core.h"
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstring>

namespace lsa
{
namespace host
{
constexpr unsigned kChainMax = 6;  // ICP iterations of one loop enqueued at once behind links (lsa_icp_link: 7 blocks, 7 mailboxes)
namespace
{
#define ICP_TRACE(...) do { if (lsa_icp_trace_on()) std::fprintf(stderr, __VA_ARGS__); } while (0)
struct Tick
{
  std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
  double Stop() const { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); }
};
inline double StampToSec(uint64_t us) { return us * 1e-6; }  // Utils::PclStampToSec
}  // namespace

static long long SteadyNs() { return std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
HostWorker::HostWorker() : T([this] { Run(); }) {}
HostWorker::~HostWorker()
{
  {
    std::lock_guard<std::mutex> l(M);
    Quit = true;
    Leaving.store(true);
  }
  Cv.notify_all();
  T.join();
}
void HostWorker::Submit(std::function<void()> job)
{
  {
    std::lock_guard<std::mutex> l(M);
    Jobs.push_back(std::move(job));
    Posted.fetch_add(1, std::memory_order_release);
  }
  Cv.notify_one();
}
void HostWorker::Expect(double seconds)
{
  PollUntil.store(SteadyNs() + static_cast<long long>(seconds * 1e9), std::memory_order_release);
  Cv.notify_one();
}
void HostWorker::Wait()
{
  std::unique_lock<std::mutex> l(M);
  Idle.wait(l, [this] { return Jobs.empty() && !Busy; });
}
void HostWorker::Run()
{
  std::unique_lock<std::mutex> l(M);
  while (true)
  {
    Cv.wait(l, [this] { return Quit || !Jobs.empty() || SteadyNs() < PollUntil.load(std::memory_order_acquire); });
    if (Jobs.empty())
    {
      if (Quit) return;  // nothing left to do
      // woken ahead of a job (Expect): poll for it, without the lock, until it is there or the time is up
      l.unlock();
      while (Posted.load(std::memory_order_acquire) == 0 && !Leaving.load() && SteadyNs() < PollUntil.load(std::memory_order_acquire))
      {
#if defined(__x86_64__)
        __builtin_ia32_pause();
#endif
      }
      PollUntil.store(0, std::memory_order_release);
      l.lock();
      continue;
    }
    std::function<void()> job = std::move(Jobs.front());
    Jobs.pop_front();
    Posted.fetch_sub(1, std::memory_order_release);
    Busy = true;
    l.unlock();
    job();
    l.lock();
    Busy = false;
    if (Jobs.empty()) Idle.notify_all();
  }
}

#define LSA_TRY(call)                          \
  do                                           \
  {                                            \
    int rc__ = (call);                         \
    if (rc__ < 0) return Fail(rc__, #call);    \
  } while (0)

// Slam::Slam (Slam.cxx:143-161)
SlamCore::SlamCore(int device)
{
  ExtractParams.neighbor_width = 4;
  ExtractParams.min_distance_to_sensor = 1.5f;
  ExtractParams.min_beam_surface_angle = 10.f;
  ExtractParams.plane_sin_angle_threshold = 0.5f;
  ExtractParams.edge_sin_angle_threshold = 0.86f;
  ExtractParams.dist_to_line_threshold = 0.20f;
  ExtractParams.edge_depth_gap_threshold = 0.15f;
  ExtractParams.edge_saliency_threshold = 1.5f;
  ExtractParams.edge_intensity_gap_threshold = 50.f;
  for (int k = 0; k < 3; ++k) LocalMaps[k] = std::make_shared<RollingGrid>();
  if (const char* e = std::getenv("LSA_ICP_AHEAD")) ICPAhead = std::atoi(e);  // default of the parameter (A/B runs)
  // the plane map takes the largest insertions (tens of thousands of keypoints per keyframe): four host threads
  // keep them shorter than the ego-motion ICP they run beside ("MapAddThreads"; the map is the same for any value)
  LocalMaps[LSA_PLANE]->SetAddThreads(4);
  for (int k = 0; k < 3; ++k) LocalMaps[k]->SetVoxelResolution(10.);
  for (int k = 0; k < 3; ++k) LocalMaps[k]->SetGridSize(50);
  LocalMaps[LSA_EDGE]->SetLeafSize(0.30);
  LocalMaps[LSA_PLANE]->SetLeafSize(0.60);
  LocalMaps[LSA_BLOB]->SetLeafSize(0.30);
  const int rc = lsa_ctx_create(device, &Ctx);
  if (rc != LSA_OK)
  {
    Ctx = nullptr;
    LastError = rc == LSA_E_NO_DEVICE ? "no usable HIP device (there is no CPU fallback)" : "lsa_ctx_create failed";
  }
  else
    for (int k = 0; k < 3; ++k)
    {
      lsa_ctx* ctx = Ctx;
      if (lsa_device_grid_create(ctx, &DevMaps[k]) == LSA_OK)
      {
        // the other parameters start from the same defaults as RollingGrid's (RollingGrid.h:170-212) and follow the same
        // setters afterwards (SetVoxelResolution snaps to the leaf size of the moment on both sides, RollingGrid.cxx:73-88)
        lsa_device_grid_set(DevMaps[k], "LeafSize", LocalMaps[k]->GetLeafSize());
      }
      else DevMaps[k] = nullptr;
      LocalMaps[k]->SetSubMapStorage([ctx, k](std::size_t n) { return lsa_target_staging(ctx, LSA_TARGET_MAP, k, static_cast<int>(n)); });
    }
  Reset();
}

SlamCore::~SlamCore()
{
  if (std::getenv("LSA_STAGE_DEBUG") && DbgFrames > 0)
    std::fprintf(stderr, "[stage debug] per frame, us: garbage+adopt %.1f | wait for the look-ahead thread %.1f | ego-motion before its loop %.1f | after the maps %.1f | registration error %.1f | localization outside its stage timers %.1f | of the first: garbage %.1f, up to the hand-over %.1f (%ld frames)\n",
                 1e6 * DbgAcc[0] / DbgFrames, 1e6 * DbgAcc[1] / DbgFrames, 1e6 * DbgAcc[2] / DbgFrames, 1e6 * DbgAcc[3] / DbgFrames, 1e6 * DbgAcc[4] / DbgFrames, 1e6 * DbgAcc[5] / DbgFrames, 1e6 * DbgAcc[6] / DbgFrames, 1e6 * DbgAcc[7] / DbgFrames, DbgFrames);
  WaitMaps();
  for (auto* g : DevMaps)
    if (g) lsa_device_grid_destroy(g);
  if (Ctx) lsa_ctx_destroy(Ctx);
}

namespace
{
// Slam::AddFrames with several frames keeps them in the last slots of the context's frame store
constexpr int kMaxDeviceFrames = 16;
constexpr int kFirstDeviceFrameSlot = 65536 - kMaxDeviceFrames - 1;

lsa_extract_params_t DefaultExtractParams()
{
  return lsa_extract_params_t{4, 1.5f, 10.f, 0.5f, 0.86f, 0.20f, 0.15f, 1.5f, 50.f};  // SSKE.h:125-148
}
}  // namespace

int SlamCore::Fail(int rc, const char* where)
{
  LastError = std::string(where) + ": " + (Ctx ? lsa_last_error(Ctx) : "no context");
  return rc;
}

// Slam::Reset (Slam.cxx:164-210)
void SlamCore::Reset(bool resetLog)
{
  WaitMaps();
  for (int k = 0; k < 3; ++k) LocalMaps[k]->Reset();
  for (auto* g : DevMaps)
    if (g) lsa_device_grid_reset(g, nullptr);
  KfLastPose = Pose::Identity();
  KfCounter = 0;
  Tworld = PreviousTworld = Trelative = Pose::Identity();
  Motion = WithinFrameMotion();
  LocalizationUncertainty = RegistrationError();
  CurrentStamp = 0;  // the "previous frame" after a reset is an empty cloud with stamp 0
  HaveFrame = false;
  // nothing announced ahead of the reset (HintNextFrame) is taken over after it, no ICP iteration stays enqueued
  NextFrameHinted = false;
  if (Ctx)
  {
    (void)lsa_upload_frame_forget(Ctx);
    (void)lsa_icp_abandon(Ctx);
  }
  if (Ctx)
    for (int s = 0; s < 3; ++s)
      for (int k = 0; k < 3; ++k) lsa_set_keypoints(Ctx, s, k, nullptr, 0);
  for (int k = 0; k < 3; ++k) { EgoDebug[k] = MatchDebug(); LocDebug[k] = MatchDebug(); KeypointCounts[k] = 0; SpecBuilt[k] = false; }
  SpecPending = false;
  for (int k = 0; k < 3; ++k)
  {
    SpecDone[k].store(false);
    SpecStaged[k] = false;
    if (Ctx) (void)lsa_drop_target_ahead(Ctx, LSA_TARGET_MAP, k);  // nothing handed over ahead of time survives a reset
  }
  CurrentFrames.clear();
  for (int k = 0; k < 3; ++k) EgoMatchSerial[k] = LocMatchSerial[k] = 0;
  if (resetLog)
  {
    NbrFrameProcessed = 0;
    LogTrajectory.clear();  // LogCovariances is left alone, as in the reference (Slam.cxx:200-209)
  }
}

Pose SlamCore::GetWorldTransform(double* time) const
{
  if (time) *time = LogTrajectory.empty() ? 0. : LogTrajectory.back().time;
  return LogTrajectory.empty() ? Pose::Identity() : LogTrajectory.back().pose;
}

// Slam::GetLatencyCompensatedWorldTransform (Slam.cxx:555-590)
Pose SlamCore::GetLatencyCompensatedWorldTransform(double* time) const
{
  const size_t n = LogTrajectory.size();
  if (time) *time = n ? LogTrajectory.back().time : 0.;
  if (n == 0) return Pose::Identity();
  if (n == 1) return LogTrajectory.back().pose;
  const StampedPose& previous = LogTrajectory[n - 2];
  const StampedPose& current = LogTrajectory[n - 1];
  // timestamps undefined or too close / extrapolation too far: the current pose
  if (std::abs(current.time - previous.time) < 1e-6) return current.pose;
  if (std::abs(Latency / (current.time - previous.time)) > MaxExtrapolationRatio) return current.pose;
  return LinearInterpolation(previous.pose, current.pose, current.time + Latency, previous.time, current.time);
}

// Slam::SetWorldTransformFromGuess (Slam.cxx:490-501)
int SlamCore::SetWorldTransformFromGuess(const Pose& guess)
{
  if (!Ctx) return LSA_E_NO_DEVICE;
  Tworld = guess;
  // the ego-motion extrapolation then gives identity, and without current keypoints the next frame skips the
  // ego-motion registration
  PreviousTworld = Tworld;
  for (int k = 0; k < 3; ++k) LSA_TRY(lsa_set_keypoints(Ctx, LSA_SET_RAW_CURRENT, k, nullptr, 0));
  return LSA_OK;
}

// Slam::GetDebugInformation (Slam.cxx:610-633)
int SlamCore::GetDebugInformation(double out[10])
{
  if (!Ctx) return LSA_E_NO_DEVICE;
  for (int i = 0; i < 10; ++i) out[i] = 0.;
  int hist[LSA_MATCH_NSTATUS];
  for (int k = 0; k < 3; ++k)
  {
    // MatchingResults::NbMatches() = RejectionsHistogram[SUCCESS] of the last iteration's match
    if (k < 2 && EgoMatchSerial[k] > 0)
    {
      LSA_TRY(lsa_match_histogram(Ctx, k, EgoMatchSerial[k], hist));
      out[k] = hist[LSA_MATCH_SUCCESS];
    }
    if (LocMatchSerial[k] > 0)
    {
      LSA_TRY(lsa_match_histogram(Ctx, k, LocMatchSerial[k], hist));
      out[2 + k] = hist[LSA_MATCH_SUCCESS];
    }
  }
  out[5] = LocalizationUncertainty.PositionError;
  out[6] = LocalizationUncertainty.OrientationError;
  out[7] = OverlapEstimation;
  out[8] = ComplyMotionLimits ? 1. : 0.;
  out[9] = Latency;
  return LSA_OK;
}

// Slam::AddFrames (Slam.cxx:230-344) + CheckFrames (:709-743)
int SlamCore::AddFrame(const lsa_point_t* pts, int n, uint64_t stampUs, uint32_t)
{
  if (!Ctx) return LSA_E_NO_DEVICE;
  Tick total;
  Stats = FrameStats();
  (void)lsa_collect_garbage(Ctx);  // buffers outgrown during the last frame (nothing on the device waits for the host here)
  DbgAcc[6] += total.Stop();
  if (!pts || n <= 0) { LastError = "SLAM input only contains empty pointclouds : exiting."; return LSA_OK; }
  if (stampUs == CurrentStamp) { LastError = "SLAM frames have the same timestamp as previous ones : frames ignored."; return LSA_OK; }
  if (pts[0].device_id != 0 && (!OtherExtractors.empty() || !OtherBaseToLidarOffsets.empty()))
  {
    // a frame of another device that has an extractor or an offset of its own
    const InputFrame f{pts, n, stampUs, 0};
    return AddFrames(&f, 1);
  }
  CurrentFrames.clear();
  DbgAcc[7] += total.Stop();
  {
    // the cloud may have been announced and uploaded ahead (HintNextFrame): it is taken over, otherwise copied now
    const int adopted = lsa_upload_frame_adopt(Ctx, pts, n);
    if (adopted < 0) return Fail(adopted, "lsa_upload_frame_adopt");
    if (adopted == 0) LSA_TRY(lsa_upload_frame(Ctx, pts, n));
  }
  if (NbrFrameProcessed == 8) { for (double& d : DbgAcc) d = 0.; DbgFrames = 0; }  // (the first frames allocate)
  DbgAcc[0] += total.Stop();
  DbgFrames++;
  int rc = ProcessCurrentFrame(stampUs);
  Latency = Stats.total = total.Stop();
  return rc;
}

int SlamCore::AddStoredFrame(int slot, uint64_t stampUs, uint32_t)
{
  if (!Ctx) return LSA_E_NO_DEVICE;
  Tick total;
  Stats = FrameStats();
  if (stampUs == CurrentStamp) { LastError = "SLAM frames have the same timestamp as previous ones : frames ignored."; return LSA_OK; }
  CurrentFrames.clear();
  LSA_TRY(lsa_frame_store_use(Ctx, slot));
  int rc = ProcessCurrentFrame(stampUs);
  Latency = Stats.total = total.Stop();
  return rc;
}

int SlamCore::HintNextFrame(const lsa_point_t* pts, int n)
{
  if (!Ctx) return LSA_E_NO_DEVICE;
  if (!pts || n <= 0) return LSA_OK;
  NextFrameHinted = false;
  LSA_TRY(lsa_upload_frame_begin(Ctx, pts, n));
  NextFrameHinted = true;
  return LSA_OK;
}

// The look-ahead extraction of the cloud announced with HintNextFrame starts as soon as (i) the current frame's own
// extraction is over and (ii) the uploader thread has enqueued the DMA: asked for after the extraction and between
// the steps of the ICP loops.
int SlamCore::TryStartLookahead()
{
  // (not while the sub-maps ahead of time are still being enqueued on the same stream: ProcessCurrentFrame)
  if (!NextFrameHinted || HoldLookahead || DevSpecRunning.load(std::memory_order_acquire) || !lsa_upload_frame_ready(Ctx)) return LSA_OK;
  NextFrameHinted = false;
  if (lsa_extract_prefetch_uploaded(Ctx, &ExtractParams) != LSA_OK) LastError = std::string("look-ahead ignored: ") + lsa_last_error(Ctx);
  return LSA_OK;
}

// The next frame's extraction (a dozen launches on the look-ahead stream once its upload has arrived) is enqueued by this
// thread between the launch of a solve and the wait for its result, not between the match and the solve: the solve's
// kernel then follows the model fits at once (it came 56 us late whenever the look-ahead was enqueued in front of it).
void SlamCore::ArmLookaheadInterlude()
{
  InterludeRan = false;
  InterludeStatus = LSA_OK;
  if (!DeviceLM) return;  // the host-driven loop has no such gap: FinishLookaheadInterlude does the work
  lsa_solve_device_interlude(Ctx, [](void* self) {
    SlamCore* core = static_cast<SlamCore*>(self);
    core->InterludeStatus = core->InterludeWork();
    core->InterludeRan = true;
  }, this);
}
int SlamCore::FinishLookaheadInterlude()
{
  if (DeviceLM) lsa_solve_device_interlude(Ctx, nullptr, nullptr);  // a solve that never got to its launch leaves it armed
  if (!InterludeRan) InterludeStatus = InterludeWork();
  InterludeRan = true;
  return InterludeStatus;
}
// what this thread enqueues while a solve runs: the next frame's extraction once its upload is there, and (once per frame,
// with the first solve of the ego-motion ICP) the sub-maps for the predicted boxes on the device
int SlamCore::InterludeWork()
{
  LSA_TRY(TryStartLookahead());
  return LSA_OK;
}

// Slam::AddFrames with several frames, one per LiDAR device (Slam.cxx:230-344; CheckFrames :709-743)
int SlamCore::AddFrames(const InputFrame* frames, int nframes)
{
  if (!Ctx) return LSA_E_NO_DEVICE;
  if (!frames || nframes <= 0 || nframes > kMaxDeviceFrames) { LastError = "AddFrames: between 1 and 16 frames"; return LSA_E_ARG; }
  // one frame of the default device with nothing configured per device: the plain path (no device-resident copy)
  const bool perDevice = !OtherExtractors.empty() || !OtherBaseToLidarOffsets.empty();
  if (nframes == 1 && (!frames[0].pts || frames[0].n <= 0 || frames[0].pts[0].device_id == 0 || !perDevice))
    return AddFrame(frames[0].pts, frames[0].n, frames[0].stampUs, frames[0].seq);
  Tick total;
  Stats = FrameStats();
  (void)lsa_collect_garbage(Ctx);
  bool allEmpty = true;
  for (int i = 0; i < nframes; ++i)
    if (frames[i].pts && frames[i].n > 0) allEmpty = false;
  if (allEmpty) { LastError = "SLAM input only contains empty pointclouds : exiting."; return LSA_OK; }
  if (frames[0].stampUs == CurrentStamp) { LastError = "SLAM frames have the same timestamp as previous ones : frames ignored."; return LSA_OK; }
  // the frames become device-resident: the extraction, the registered frame and the overlap estimator read them
  CurrentFrames.clear();
  Device0AzimuthalResolution = lsa_get_azimuthal_resolution(Ctx);
  for (int i = 0; i < nframes; ++i)
  {
    if (!frames[i].pts || frames[i].n <= 0) continue;  // "SLAM input frame i is an empty pointcloud : frame ignored."
    const int device = frames[i].pts[0].device_id;
    // the azimuthal resolution is estimated from a device's first frame while it is handed over
    float* az = device == 0 ? &Device0AzimuthalResolution : (OtherExtractors.count(device) ? &OtherExtractors[device].azimuthalResolution : nullptr);
    if (!az && OtherExtractors.empty()) az = &Device0AzimuthalResolution;  // the default extractor stands in (Slam.cxx:766-772)
    lsa_set_azimuthal_resolution(Ctx, az ? *az : 1.f);
    const int slot = kFirstDeviceFrameSlot + i;
    if (lsa_frame_store_put(Ctx, slot, frames[i].pts, frames[i].n) != LSA_OK)
    {
      lsa_set_azimuthal_resolution(Ctx, Device0AzimuthalResolution);  // device 0's value goes back where it lives
      CurrentFrames.clear();
      return Fail(LSA_E_HIP, "AddFrames");
    }
    if (az) *az = lsa_get_azimuthal_resolution(Ctx);
    CurrentFrames.push_back({slot, frames[i].n, device, StampToSec(frames[i].stampUs) - StampToSec(frames[0].stampUs)});
  }
  lsa_set_azimuthal_resolution(Ctx, Device0AzimuthalResolution);
  int rc = ProcessCurrentFrame(frames[0].stampUs);
  Latency = Stats.total = total.Stop();
  return rc;
}

int SlamCore::SetExtractorParam(int deviceId, const std::string& name, double v)
{
  if (deviceId < 0 || deviceId > 255) return LSA_E_ARG;
  if (deviceId == 0) return SetParam(name, v);
  const bool known = OtherExtractors.count(deviceId) != 0;
  DeviceExtractor e = known ? OtherExtractors[deviceId] : DeviceExtractor{DefaultExtractParams(), 0.f};
  lsa_extract_params_t& p = e.params;
  if (name == "NeighborWidth") p.neighbor_width = static_cast<int>(v);
  else if (name == "MinDistanceToSensor") p.min_distance_to_sensor = static_cast<float>(v);
  else if (name == "MinBeamSurfaceAngle") p.min_beam_surface_angle = static_cast<float>(v);
  else if (name == "PlaneSinAngleThreshold") p.plane_sin_angle_threshold = static_cast<float>(v);
  else if (name == "EdgeSinAngleThreshold") p.edge_sin_angle_threshold = static_cast<float>(v);
  else if (name == "EdgeDepthGapThreshold") p.edge_depth_gap_threshold = static_cast<float>(v);
  else if (name == "EdgeSaliencyThreshold") p.edge_saliency_threshold = static_cast<float>(v);
  else if (name == "EdgeIntensityGapThreshold") p.edge_intensity_gap_threshold = static_cast<float>(v);
  else if (name == "AzimuthalResolution") e.azimuthalResolution = static_cast<float>(v);
  else { LastError = "unknown extractor parameter " + name; return LSA_E_ARG; }
  OtherExtractors[deviceId] = e;
  return LSA_OK;
}

int SlamCore::GetExtractorParam(int deviceId, const std::string& name, double* v) const
{
  if (!v) return LSA_E_ARG;
  if (deviceId == 0) return GetParam(name, v);
  const auto it = OtherExtractors.find(deviceId);
  if (it == OtherExtractors.end()) return LSA_E_ARG;
  const lsa_extract_params_t& p = it->second.params;
  if (name == "NeighborWidth") *v = p.neighbor_width;
  else if (name == "MinDistanceToSensor") *v = p.min_distance_to_sensor;
  else if (name == "MinBeamSurfaceAngle") *v = p.min_beam_surface_angle;
  else if (name == "PlaneSinAngleThreshold") *v = p.plane_sin_angle_threshold;
  else if (name == "EdgeSinAngleThreshold") *v = p.edge_sin_angle_threshold;
  else if (name == "EdgeDepthGapThreshold") *v = p.edge_depth_gap_threshold;
  else if (name == "EdgeSaliencyThreshold") *v = p.edge_saliency_threshold;
  else if (name == "EdgeIntensityGapThreshold") *v = p.edge_intensity_gap_threshold;
  else if (name == "AzimuthalResolution") *v = it->second.azimuthalResolution;
  else return LSA_E_ARG;
  return LSA_OK;
}

int SlamCore::SetBaseToLidarOffset(int deviceId, const Pose& offset)
{
  if (deviceId < 0 || deviceId > 255) return LSA_E_ARG;
  if (deviceId == 0) BaseToLidarOffset = offset;
  else OtherBaseToLidarOffsets[deviceId] = offset;
  return LSA_OK;
}

// Slam::GetBaseToLidarOffset (Slam.cxx): identity for a device nobody configured
Pose SlamCore::GetBaseToLidarOffset(int deviceId) const
{
  if (deviceId == 0) return BaseToLidarOffset;
  const auto it = OtherBaseToLidarOffsets.find(deviceId);
  return it == OtherBaseToLidarOffsets.end() ? Pose::Identity() : it->second;
}

int SlamCore::ProcessCurrentFrame(uint64_t stampUs)
{
  CurrentStamp = stampUs;
  CurrentTime = StampToSec(stampUs);
  HaveFrame = true;
  {
    Tick t;
    AheadWorker.Wait();  // it reads the previous frame's keypoint buffers, which the extraction is about to rotate (long done)
    DbgAcc[1] += t.Stop();
  }
  // The look-ahead stream carries, in this order of urgency: the previous keyframe's insertion and the sub-maps extracted
  // ahead of time (this frame's localization waits for them), the next frame's ego-motion targets, the next frame's
  // extraction (nobody waits for it before the next AddFrame).  An upload that is there early must not put the extraction in
  // front: with the cloud staged by four threads the sub-map stage went from 0.02 to 0.12 ms per frame (1 250 -> 1 100 frames/s).
  // Held back until the speculation has been enqueued (BeginSubMapSpeculation, DevSpecRunning).
  HoldLookahead = DeviceMapsInUse() && SubMapsAhead && MapUpdate != MappingMode::NONE;
  lsa_set_knn_lanes(Ctx, LSA_EDGE, KnnLanesEdges);
  lsa_set_knn_lanes(Ctx, LSA_PLANE, KnnLanesPlanes);
  lsa_set_knn_lanes(Ctx, LSA_BLOB, KnnLanesBlobs);
  lsa_set_knn_rounds(Ctx, LSA_EDGE, KnnRoundsEdges);
  lsa_set_knn_rounds(Ctx, LSA_PLANE, KnnRoundsPlanes);
  lsa_set_knn_rounds(Ctx, LSA_BLOB, KnnRoundsBlobs);
  lsa_set_fused_match(Ctx, FusedMatch ? 1 : 0);
  {
    Tick t;
    int rc = ExtractKeypoints();
    if (rc < 0) return rc;
    Stats.extract = t.Stop();
  }
  int rc = ComputeEgoMotion();
  if (rc < 0) return rc;
  rc = Localization();
  if (rc < 0) return rc;
  Tick tail;
  // confidence estimators: before the maps update, which invalidates the sub-maps (Slam.cxx:269-280)
  if (OverlapSamplingRatio > 0)
  {
    rc = EstimateOverlap();
    if (rc < 0) return rc;
  }
  if (TimeWindowDuration > 0) CheckMotionLimits();
  if (MapUpdate == MappingMode::ADD_KPTS_TO_FIXED_MAP || MapUpdate == MappingMode::UPDATE)
  {
    Tick t;
    rc = UpdateMapsUsingTworld();
    if (rc < 0) return rc;
    Stats.maps = t.Stop();
  }
  // the words the next frame's localization reduces its keypoints' boxes into, armed while the device has nothing to do
  if (LocalizationStartFused && DeviceMapsInUse() && MapUpdate != MappingMode::NONE) LSA_TRY(lsa_arm_localization_boxes(Ctx));
  LogCurrentFrameState(CurrentTime);
  // the next frame's first jobs for the look-ahead thread (the sub-maps under the predicted pose, the ego-motion targets) come
  // within a fraction of a millisecond when the caller replays: the thread is awake for them
  if (WorkerPrewake && DeviceMapsInUse()) AheadWorker.Expect(300e-6);
  NbrFrameProcessed++;
  DbgAcc[3] += tail.Stop() - Stats.maps;
  return LSA_OK;
}

// Slam::ExtractKeypoints (Slam.cxx:746-810)
int SlamCore::ExtractKeypoints()
{
  int counts[3];
  unsigned mask = 0;
  for (int k = 0; k < 3; ++k)
    if (UseKeypoints[k]) mask |= 1u << k;
  lsa_set_keypoint_types(Ctx, mask);  // unused types come out empty (Slam.cxx:789-793)
  if (!CurrentFrames.empty()) return ExtractFrames();
  LSA_TRY(lsa_extract_keypoints(Ctx, &ExtractParams, counts));  // also: PreviousRawKeypoints = CurrentRawKeypoints
  for (int k = 0; k < 3; ++k) KeypointCounts[k] = counts[k];
  // AggregateFrames(keypoints, false): LIDAR -> BASE (Slam.cxx:1551-1573); skipped when identity
  if (!IsApprox(BaseToLidarOffset, Pose::Identity()))
    for (int k = 0; k < 3; ++k)
      if (counts[k] > 0) LSA_TRY(lsa_transform_keypoints(Ctx, LSA_SET_RAW_CURRENT, k, BaseToLidarOffset.m, 0.));
  if (NextStoredSlot >= 0)
  {
    // look-ahead: the next frame's extraction runs on its own stream beside this frame's registration; a slot that
    // does not exist is reported and otherwise ignored
    if (lsa_extract_prefetch(Ctx, NextStoredSlot, &ExtractParams) != LSA_OK) LastError = std::string("look-ahead ignored: ") + lsa_last_error(Ctx);
    NextStoredSlot = -1;
  }
  LSA_TRY(TryStartLookahead());
  return LSA_OK;
}

// Slam::ExtractKeypoints with several device frames (Slam.cxx:753-801) + AggregateFrames(keypoints, false) (:1512-1578)
int SlamCore::ExtractFrames()
{
  for (int k = 0; k < 3; ++k) KeypointCounts[k] = 0;
  bool first = true;
  for (const HeldFrame& f : CurrentFrames)
  {
    // the extractor of the frame's device; with a single extractor configured, that one stands in (Slam.cxx:762-779)
    const lsa_extract_params_t* params = nullptr;
    float* az = nullptr;
    if (f.device == 0) { params = &ExtractParams; az = &Device0AzimuthalResolution; }
    else if (OtherExtractors.count(f.device)) { params = &OtherExtractors[f.device].params; az = &OtherExtractors[f.device].azimuthalResolution; }
    else if (OtherExtractors.empty()) { params = &ExtractParams; az = &Device0AzimuthalResolution; }
    else
    {
      LastError = "Input frame comes from LiDAR device " + std::to_string(f.device) + " but no keypoints extractor has been set for this device : ignoring frame.";
      continue;
    }
    LSA_TRY(lsa_frame_store_use(Ctx, f.slot));
    lsa_set_azimuthal_resolution(Ctx, *az);
    // the offset is looked up by the device the points say they come from (Slam.cxx:1540)
    const Pose offset = GetBaseToLidarOffset(f.device);
    const bool identity = IsApprox(offset, Pose::Identity());
    int counts[3] = {0, 0, 0};
    if (first)
    {
      LSA_TRY(lsa_extract_keypoints(Ctx, params, counts));  // also: PreviousRawKeypoints = CurrentRawKeypoints
      if (!identity || f.timeOffset != 0.)
        for (int k = 0; k < 3; ++k)
          if (counts[k] > 0) LSA_TRY(lsa_transform_keypoints(Ctx, LSA_SET_RAW_CURRENT, k, offset.m, f.timeOffset));
    }
    else
      LSA_TRY(lsa_extract_keypoints_more(Ctx, params, identity ? nullptr : offset.m, f.timeOffset, counts));
    first = false;
    for (int k = 0; k < 3; ++k) KeypointCounts[k] += counts[k];
  }
  lsa_set_azimuthal_resolution(Ctx, Device0AzimuthalResolution);
  if (first)
    // no frame had an extractor: the current keypoints are empty (the previous ones are not consulted then)
    for (int k = 0; k < 3; ++k) LSA_TRY(lsa_set_keypoints(Ctx, LSA_SET_RAW_CURRENT, k, nullptr, 0));
  return LSA_OK;
}

// The kd-trees ComputeEgoMotion builds on the previous frame's keypoints (Slam.cxx:845-860) only depend on this
// frame's keypoints: their device counterparts are built now, beside this frame's registration, and found ready by
// the next frame.
int SlamCore::PrepareNextEgoMotionTargets()
{
  if (!BuildTargetsAhead) return LSA_OK;
  if (!(EgoMotion == EgoMotionMode::REGISTRATION || EgoMotion == EgoMotionMode::MOTION_EXTRAPOLATION_AND_REGISTRATION)) return LSA_OK;
  return lsa_prepare_previous_targets(Ctx, (1u << LSA_EDGE) | (1u << LSA_PLANE));
}

lsa_match_params_t SlamCore::EgoMatchParams() const
{
  lsa_match_params_t p;
  std::memset(&p, 0, sizeof(p));
  p.single_edge_per_ring = 1;  // Slam.cxx:879
  p.max_neighbors_distance = EgoMotionMaxNeighborsDistance;
  p.edge_nb_neighbors = EgoMotionEdgeNbNeighbors;
  p.edge_min_nb_neighbors = EgoMotionEdgeMinNbNeighbors;
  p.edge_max_model_error = EgoMotionEdgeMaxModelError;
  p.plane_nb_neighbors = EgoMotionPlaneNbNeighbors;
  p.planarity_threshold = EgoMotionPlanarityThreshold;
  p.plane_max_model_error = EgoMotionPlaneMaxModelError;
  p.blob_nb_neighbors = 10;
  p.saturation_distance = 1.;
  return p;
}
lsa_match_params_t SlamCore::LocMatchParams() const
{
  lsa_match_params_t p;
  std::memset(&p, 0, sizeof(p));
  p.single_edge_per_ring = 0;  // Slam.cxx:1057
  p.max_neighbors_distance = LocalizationMaxNeighborsDistance;
  p.edge_nb_neighbors = LocalizationEdgeNbNeighbors;
  p.edge_min_nb_neighbors = LocalizationEdgeMinNbNeighbors;
  p.edge_max_model_error = LocalizationEdgeMaxModelError;
  p.plane_nb_neighbors = LocalizationPlaneNbNeighbors;
  p.planarity_threshold = LocalizationPlanarityThreshold;
  p.plane_max_model_error = LocalizationPlaneMaxModelError;
  p.blob_nb_neighbors = LocalizationBlobNbNeighbors;
  p.saturation_distance = 1.;
  return p;
}

// Slam::ComputeEgoMotion (Slam.cxx:813-972)
int SlamCore::ComputeEgoMotion()
{
  Tick tpre;
  Trelative = Pose::Identity();
  if (AheadStatus < 0) { const int rc = AheadStatus; AheadStatus = 0; return Fail(rc, "lsa_prepare_previous_targets (look-ahead thread)"); }
  if (LogTrajectory.size() >= 2 &&
      (EgoMotion == EgoMotionMode::MOTION_EXTRAPOLATION || EgoMotion == EgoMotionMode::MOTION_EXTRAPOLATION_AND_REGISTRATION))
  {
    const double t = StampToSec(CurrentStamp);
    const double t1 = LogTrajectory[LogTrajectory.size() - 1].time;
    const double t0 = LogTrajectory[LogTrajectory.size() - 2].time;
    if (!(std::abs((t - t1) / (t1 - t0)) > MaxExtrapolationRatio))
    {
      const Pose next = LinearInterpolation(PreviousTworld, Tworld, t, t0, t1);
      Trelative = Inverse(Tworld) * next;
    }
  }
  const bool registers = EgoMotion == EgoMotionMode::REGISTRATION || EgoMotion == EgoMotionMode::MOTION_EXTRAPOLATION_AND_REGISTRATION;
  {
    const int rc = BeginSubMapSpeculation(Tworld * Trelative);
    if (rc < 0) return rc;
    HoldLookahead = false;  // (from here on DevSpecRunning says whether the speculation is still being enqueued)
  }
  if (!registers) return LSA_OK;

  // kd-trees on the previous frame's raw keypoints -> device search grids, no PCIe traffic
  for (int k : {LSA_EDGE, LSA_PLANE})
  {
    lsa_set_target_cell_size(Ctx, LSA_TARGET_PREVIOUS, k, static_cast<float>(k == LSA_EDGE ? KnnCellSizeEgoMotionEdges : KnnCellSizeEgoMotion));
    LSA_TRY(lsa_set_target_from_set(Ctx, LSA_TARGET_PREVIOUS, k, LSA_SET_RAW_PREVIOUS));
  }
  TotalMatchedKeypoints = 0;
  lsa_match_params_t mp = EgoMatchParams();
  const unsigned mask = (1u << LSA_EDGE) | (1u << LSA_PLANE);
  auto saturation = [&](unsigned icpIter) {
    const double iterRatio = icpIter / static_cast<double>(EgoMotionICPMaxIter - 1);
    return (1 - iterRatio) * EgoMotionInitSaturationDistance + iterRatio * EgoMotionFinalSaturationDistance;
  };
  auto configure = [&](LocalOptimizer& optimizer) {
    optimizer.SetDeviceLoop(DeviceLM);
    optimizer.SetTwoDMode(TwoDMode);
    optimizer.SetLMMaxIter(EgoMotionLMMaxIter);
    optimizer.SetMinMatches(MinNbMatchedKeypoints);
    optimizer.UseDeviceResiduals(mask);
  };
  // Iteration i + 1 is enqueued behind a gate (lsa_icp_gate) while iteration i runs: when the solve's result arrives its
  // launches are in the queue already, and all that is between the solve and the next search is one store the gate polls
  // for (or the call that calls them off: Slam.cxx:919-923, 950).
  // (With the maps on the HOST this loop stays as it was.  Between its iterations this thread then hands sub-maps the map
  // workers have extracted to the look-ahead stream (StageSpeculativeSubMaps: a copy out of pinned memory and a grid build).
  // With a gate waiting on the registration's stream at that moment the device delivered no result for two seconds in some
  // assignments of the streams to the hardware queues (seen with a second context alive in the process), the solve was then
  // redone on the host with the NEXT iteration's saturation distance already in force: 1.5e-5 m beside the oracle.  The
  // hang is not understood (LSA_ICP_AHEAD_HOSTMAPS=1 + LSA_ICP_TRACE=1 reproduce it); the fall-back's wrong distance is
  // fixed (the solve on the host takes the distances the device solve was enqueued with).)
  static const bool aheadWithHostMaps = std::getenv("LSA_ICP_AHEAD_HOSTMAPS") != nullptr;  // (diagnostics)
  static const int aheadLoops = std::getenv("LSA_ICP_AHEAD_LOOPS") ? std::atoi(std::getenv("LSA_ICP_AHEAD_LOOPS")) : 3;  // (diagnostics: 1 ego-motion only, 2 localization only)
  // ICPAhead = 2: the WHOLE loop is enqueued at once, every solve leaving pose and start point for the iteration behind it
  // on the device (lsa_icp_link) -- no gate, no host between two iterations, nothing spins; this thread reads the results as
  // they arrive, takes the decisions the device has taken already and does its own pose algebra beside the running device.
  bool chain = ICPAhead >= 2 && DeviceLM && FusedMatch && EgoMotionICPMaxIter <= kChainMax && (aheadLoops & 1);
  bool ahead = !chain && ICPAhead >= 1 && DeviceLM && FusedMatch && (DeviceMapsInUse() || MapUpdate == MappingMode::NONE || aheadWithHostMaps) && (aheadLoops & 1);
  if (ahead || chain) lsa_icp_abandon(Ctx);
  unsigned chained = 0;            // iterations 0 .. chained - 1 are in the queue (links)
  long long chainSerial[kChainMax][3] = {};
  bool enqueued = false;           // this iteration's launches are in the queue (their gate has been answered)
  long long aheadSerial[3] = {0, 0, 0};
  auto callOff = [&](int& ticket) {
    if (ticket < 0) return;
    lsa_icp_cancel(Ctx, ticket);
    lsa_solve_device_drop(Ctx);
    ticket = -1;
  };

  DbgAcc[2] += tpre.Stop();
  for (unsigned icpIter = 0; icpIter < EgoMotionICPMaxIter; ++icpIter)
  {
    Tick ticp;
    mp.saturation_distance = saturation(icpIter);
    LocalOptimizer optimizer(Ctx);
    configure(optimizer);
    optimizer.SetPosePrior(Trelative);
    bool begun = enqueued;  // the solve of this iteration is in flight
    if (icpIter < chained)
    {
      begun = true;
      for (int k : {LSA_EDGE, LSA_PLANE}) EgoMatchSerial[k] = chainSerial[icpIter][k];
    }
    else if (!enqueued)
    {
      // both keypoint types are matched concurrently and nothing is read back: the number of matches
      // arrives with the optimizer's first evaluation
      LSA_TRY(lsa_match_types(Ctx, LSA_TARGET_PREVIOUS, mask, LSA_SET_RAW_CURRENT, &mp, Trelative.m, nullptr));
      for (int k : {LSA_EDGE, LSA_PLANE}) EgoMatchSerial[k] = lsa_match_serial(Ctx, k);
      if (chain)
      {
        // this iteration's solve and every iteration behind it, as far as they can ride behind links
        double prior[6];
        ToXYZRPY(Trelative, prior);
        lsa_icp_link_t link;
        std::memset(&link, 0, sizeof(link));
        for (unsigned j = icpIter; j < EgoMotionICPMaxIter; ++j)
        {
          int leave = j + 1 < EgoMotionICPMaxIter ? lsa_icp_link(Ctx) : -1;
          if (leave < 0) leave = -1;
          link.first = j == icpIter ? 1 : 0;
          const int rc = lsa_solve_device_begin_linked(Ctx, mask, j == icpIter ? prior : nullptr, TwoDMode ? 1 : 0, static_cast<int>(EgoMotionLMMaxIter), static_cast<int>(MinNbMatchedKeypoints), leave,
                                                       leave >= 0 ? &link : nullptr);
          if (rc < 0) { lsa_icp_abandon(Ctx); return Fail(rc, "lsa_solve_device_begin_linked"); }
          chained = j + 1;
          if (leave < 0) break;
          lsa_match_params_t next = mp;
          next.saturation_distance = saturation(j + 1);
          const int mrc = lsa_match_types_gated(Ctx, LSA_TARGET_PREVIOUS, mask, LSA_SET_RAW_CURRENT, &next, 0);
          if (mrc < 0) { lsa_icp_abandon(Ctx); return Fail(mrc, "lsa_match_types_gated"); }
          if (mrc != 0) { lsa_icp_cancel(Ctx, leave); break; }  // this match cannot wait behind a link: the loop goes on in line from there
          for (int k : {LSA_EDGE, LSA_PLANE}) chainSerial[j + 1][k] = lsa_match_serial(Ctx, k);
        }
        chain = false;  // (enqueued once; whatever is not in the queue now runs in line)
        begun = true;
      }
      else if (ahead)
      {
        LSA_TRY(optimizer.Begin(false));
        begun = true;
      }
    }
    else
      for (int k : {LSA_EDGE, LSA_PLANE}) EgoMatchSerial[k] = aheadSerial[k];
    ICP_TRACE("[ego %u] top: ahead %d begun %d enqueued %d\n", icpIter, (int)ahead, (int)begun, (int)enqueued);
    enqueued = false;
    int ticket = -1;
    if (ahead && begun && icpIter + 1 < EgoMotionICPMaxIter)
    {
      ticket = lsa_icp_gate(Ctx);
      ICP_TRACE("[ego %u] gate ticket %d\n", icpIter, ticket);
      if (ticket < 0) { ticket = -1; ahead = false; }
      else
      {
        lsa_match_params_t next = mp;
        next.saturation_distance = saturation(icpIter + 1);
        int rc = lsa_match_types_gated(Ctx, LSA_TARGET_PREVIOUS, mask, LSA_SET_RAW_CURRENT, &next, 0);
        if (rc == 0)
        {
          for (int k : {LSA_EDGE, LSA_PLANE}) aheadSerial[k] = lsa_match_serial(Ctx, k);
          rc = lsa_solve_device_begin(Ctx, mask, nullptr, TwoDMode ? 1 : 0, static_cast<int>(EgoMotionLMMaxIter), static_cast<int>(MinNbMatchedKeypoints));
          if (rc < 0) { lsa_icp_cancel(Ctx, ticket); return Fail(rc, "lsa_solve_device_begin"); }
        }
        else
        {
          lsa_icp_cancel(Ctx, ticket);
          ticket = -1;
          ahead = false;
          if (rc < 0) return Fail(rc, "lsa_match_types_gated");
        }
      }
    }
    // the targets of the NEXT frame's ego-motion, which are this frame's keypoints, are built beside this registration:
    // enqueued (ten launches and two copies on the look-ahead stream: a host thread of their own issues them, this one
    // goes on to the solve) while the first iteration's kernels -- the solve's included -- are on their way
    if (icpIter == 0 && BuildTargetsAhead)
    {
      AheadStatus = 0;
      AheadWorker.Submit([this] { AheadStatus = PrepareNextEgoMotionTargets(); });
    }
    // while the device is busy with this iteration: sub-maps the workers have finished meanwhile go to the device
    if (!SpecPending) LSA_TRY(StageSpeculativeSubMaps());
    if (!begun) ArmLookaheadInterlude();
    Stats.ego_icp += ticp.Stop();
    Stats.ego_iters++;

    Tick tlm;
    SolveSummary summary;
    if (begun)
    {
      Tick tdbg;
      const int irc = InterludeWork();
      const double dbgInterlude = tdbg.Stop();
      int rc = optimizer.End(summary);
      ICP_TRACE("[ego %u] End rc %d irc %d ticket %d\n", icpIter, rc, irc, ticket);
      static const bool gateDebug = std::getenv("LSA_GATE_DEBUG") != nullptr;
      if (gateDebug && tdbg.Stop() > 0.01)
        std::fprintf(stderr, "[gate debug] ego iteration %u: enqueue %.3f ms, interlude %.3f ms, until the result %.3f ms, rc %d\n", icpIter, 1e3 * Stats.ego_icp, 1e3 * dbgInterlude, 1e3 * tdbg.Stop(), rc);
      if (rc == LSA_E_GATE)
      {
        // the gate gave up waiting for this thread (it was held up for 50 ms): nothing of the iteration ran.  Whatever
        // waits behind it is called off, the iteration is done again in line, the rest of the loop without gates.
        callOff(ticket);
        lsa_icp_abandon(Ctx);
        ahead = false;
        IcpGateTimeouts++;
        LSA_TRY(lsa_match_types(Ctx, LSA_TARGET_PREVIOUS, mask, LSA_SET_RAW_CURRENT, &mp, Trelative.m, nullptr));
        for (int k : {LSA_EDGE, LSA_PLANE}) EgoMatchSerial[k] = lsa_match_serial(Ctx, k);
        rc = optimizer.Solve(summary);
      }
      if (rc < 0) { callOff(ticket); lsa_icp_abandon(Ctx); return Fail(rc, "LocalOptimizer::Solve (ego-motion)"); }
      if (rc == 1) { ticket = -1; ahead = false; chained = 0; }  // solved on the host: what was enqueued ahead has been called off
      if (irc < 0) { callOff(ticket); return irc; }
    }
    else
    {
      LSA_TRY(optimizer.Solve(summary));
      LSA_TRY(FinishLookaheadInterlude());
    }
    TotalMatchedKeypoints = summary.num_matches;
    if (lsa_icp_trace_on())
      std::fprintf(stderr, "[icp] frame %u ego %u: matches %d evals %d steps %d cost %.17g -> %.17g%s\n", NbrFrameProcessed, icpIter, summary.num_matches, summary.num_evaluations,
                   summary.num_successful_steps, summary.initial_cost, summary.final_cost, summary.skipped ? " skipped" : "");
    if (SpecPending)
    {
      // the predicted bounding boxes have long arrived: the map workers extract the sub-maps from here on
      const int rc = FinishSubMapSpeculation();
      if (rc < 0) { callOff(ticket); return rc; }
    }
    Stats.ego_lm += tlm.Stop();
    Stats.lm_evals += summary.num_evaluations;
    // (links: the device has taken the same decision from the same result -- the iterations behind this one do nothing;
    //  what their matches announced on the host is taken back)
    const bool more = icpIter + 1 < chained;
    if (summary.skipped) { callOff(ticket); if (more) lsa_icp_abandon(Ctx); break; }  // "Not enough keypoints, EgoMotion skipped for this frame."
    Trelative = optimizer.GetOptimizedPose();
    if (summary.num_successful_steps == 1) { callOff(ticket); if (more) lsa_icp_abandon(Ctx); break; }
    if (ticket >= 0)
    {
      double prior[6];
      ToXYZRPY(Trelative, prior);  // LocalOptimizer::SetPosePrior of the next iteration
      const int rc = lsa_icp_post(Ctx, ticket, Trelative.m, prior, nullptr, nullptr, 0., 0.);
      if (rc < 0) return Fail(rc, "lsa_icp_post");
      enqueued = true;
    }
  }
  if (KeepMatchDebug)
    for (int k : {LSA_EDGE, LSA_PLANE})
    {
      const int n = lsa_keypoint_count(Ctx, LSA_SET_RAW_CURRENT, k);
      EgoDebug[k].status.assign(n, LSA_MATCH_UNKOWN);
      EgoDebug[k].weights.assign(n, 0.);
      if (n > 0) LSA_TRY(lsa_download_match(Ctx, k, EgoDebug[k].status.data(), EgoDebug[k].weights.data(), nullptr, n));
    }
  return LSA_OK;
}

// Slam::Localization (Slam.cxx:975-1175)
int SlamCore::Localization()
{
  Tick tloc;
  struct AtExit { SlamCore* c; Tick* t; ~AtExit() { c->DbgAcc[5] += t->Stop() - (c->Stats.loc_icp + c->Stats.loc_lm + c->Stats.submap + c->Stats.undistort); } } atExit{this, &tloc};
  PreviousTworld = Tworld;
  Tworld = PreviousTworld * Trelative;
  // With the maps on the device the reset, the first undistortion and the keypoints' boxes under the pose guess (which the
  // grids read on the device) are one launch; otherwise the three steps as the reference lists them.
  const bool oneLaunch = LocalizationStartFused && DeviceMapsInUse() && MapUpdate != MappingMode::NONE;
  bool boxesEnqueued = false;
  if (oneLaunch)
  {
    Tick t;
    Pose d0 = Pose::Identity(), d1 = Pose::Identity();
    if (Undistortion)
    {
      double t0, t1;
      LSA_TRY(lsa_keypoint_time_range(Ctx, LSA_SET_RAW_CURRENT, &t0, &t1));  // what the working keypoints are about to be
      InitUndistortion(t0, t1);
      RefineUndistortion(&d0, &d1);
    }
    LSA_TRY(lsa_localization_begin(Ctx, Undistortion ? d0.m : nullptr, Undistortion ? d1.m : nullptr, Motion.Time0, Motion.Time1, Tworld.m));
    boxesEnqueued = true;
    Stats.undistort += t.Stop();
  }
  else
  {
    LSA_TRY(lsa_reset_working_keypoints(Ctx));  // CurrentUndistortedKeypoints = CurrentRawKeypoints
    if (Undistortion)
    {
      Tick t;
      int rc = InitUndistortion();
      if (rc < 0) return rc;
      rc = RefineUndistortion();
      if (rc < 0) return rc;
      Stats.undistort += t.Stop();
    }
  }

  if (DeviceMapsInUse())
  {
    // Slam.cxx:1003-1037 with the maps on the device: a map whose last keyframe changed a point gives a new sub-map --
    // the voxels the box of the current keypoints (at the initial pose guess) touches -- written straight into the
    // target; only the box (read back once for all types) and the sub-maps' sizes cross the bus
    Tick t;
    // the workers have enqueued the previous keyframe's insertions, and the sub-maps ahead of time (NOT the next frame's
    // ego-motion targets, which the look-ahead thread may still be enqueueing: nothing here depends on them)
    for (auto& w : MapWorker) w.Wait();
    {
      // the sub-maps ahead of time: waited for -- and when that took time twice in a row (the ego-motion ICP of this sensor
      // is shorter than insertion + extraction + grid build), not tried again for a while ("SubMapsAheadAdaptive")
      Tick tw;
      while (DevSpecRunning.load(std::memory_order_acquire)) std::this_thread::yield();
      const bool tried = DevSpec[0] || DevSpec[1] || DevSpec[2];
      if (tried && SubMapsAheadAdaptive && tw.Stop() > 80e-6)
      {
        DevSpecGood = 0;
        if (++DevSpecLate >= 2)
        {
          // not tried for a while; twice as long every time it happens again soon (a sensor whose ego-motion ICP is simply
          // too short ends up trying once in 64 frames, an occasional hiccup costs four)
          DevSpecBackoff = DevSpecPenalty;
          DevSpecPenalty = std::min(2 * DevSpecPenalty, 64);
          DevSpecLate = 0;
        }
      }
      else if (tried)
      {
        DevSpecLate = 0;
        if (++DevSpecGood >= 8) DevSpecPenalty = 4;
      }
    }
    Stats.maps_wait = t.Stop();
    if (DevSpecStatus < 0) { const int rc = DevSpecStatus; DevSpecStatus = 0; return Fail(rc, "sub-maps ahead of time (look-ahead thread)"); }
    Stats.maps_async = std::max(MapJobSeconds[0], std::max(MapJobSeconds[1], MapJobSeconds[2]));
    for (int k = 0; k < 3; ++k)
      if (MapJobFailed[k]) { MapJobFailed[k] = 0; return Fail(LSA_E_HIP, "lsa_device_grid_add_staged (map worker)"); }
    bool need[3], checking[3] = {false, false, false}, any = false;
    for (int k = 0; k < 3; ++k)
    {
      need[k] = UseKeypoints[k] && !lsa_device_grid_submap_valid(DevMaps[k]);
      any = any || need[k];
    }
    // the boxes of the current keypoints under the pose guess stay on the device: the grids read them there
    if (any && MapUpdate != MappingMode::NONE && !boxesEnqueued) LSA_TRY(lsa_keypoint_bboxes_begin(Ctx, LSA_SET_WORKING, Tworld.m));
    for (int k = 0; k < 3; ++k)
    {
      if (!need[k]) continue;
      lsa_set_target_cell_size(Ctx, LSA_TARGET_MAP, k, static_cast<float>((k == LSA_EDGE ? KnnCellScaleMapsEdges : KnnCellScaleMaps) * LocalMaps[k]->GetLeafSize()));
      if (MapUpdate == MappingMode::NONE) LSA_TRY(lsa_device_grid_build_submap_begin(DevMaps[k], nullptr, nullptr, -1, LSA_TARGET_MAP, k));
      else
      {
        // extracted ahead of time for the predicted box: stands when the actual box touches the same outer voxels (the
        // comparisons of all the maps are enqueued first, then waited for)
        if (DevSpec[k] && lsa_device_grid_submap_ahead_take_begin(DevMaps[k], k, KeypointCounts[k] / 2, LSA_TARGET_MAP, k) == 1) { checking[k] = true; continue; }
        if (LocalMaps[k]->IsTimeThreshold()) LSA_TRY(lsa_device_grid_clear_old_points(DevMaps[k], CurrentTime));
        LSA_TRY(lsa_device_grid_build_submap_begin_for_keypoints(DevMaps[k], k, KeypointCounts[k] / 2, LSA_TARGET_MAP, k));
      }
    }
    for (int k = 0; k < 3; ++k)
    {
      if (!checking[k]) continue;
      int taken = 0;
      LSA_TRY(lsa_device_grid_submap_ahead_take_end(DevMaps[k], &taken));
      if (taken) { need[k] = false; Stats.submap_spec_hits++; SubMapSpecHitsTotal++; continue; }
      if (LocalMaps[k]->IsTimeThreshold()) LSA_TRY(lsa_device_grid_clear_old_points(DevMaps[k], CurrentTime));
      LSA_TRY(lsa_device_grid_build_submap_begin_for_keypoints(DevMaps[k], k, KeypointCounts[k] / 2, LSA_TARGET_MAP, k));
    }
    for (bool& b : DevSpec) b = false;
    for (int k = 0; k < 3; ++k)  // the extractions follow one another on the context's stream, their sizes come back together
      if (need[k]) LSA_TRY(lsa_device_grid_build_submap_end(DevMaps[k]));
    Stats.submap += t.Stop();
    HoldLookahead = false;  // (the extraction is enqueued from the loop's interlude, behind this frame's first search: not here, in front of it)
  }
  else
  {
    Tick t;
    if (SpecPending)
    {
      const int rc = FinishSubMapSpeculation();
      if (rc < 0) return rc;
    }
    WaitMaps();  // the previous keyframe's insertion and the sub-maps extracted ahead of time
    Stats.maps_wait = t.Stop();
    Stats.maps_async = std::max(MapJobSeconds[0], std::max(MapJobSeconds[1], MapJobSeconds[2]));
    // which sub-maps are new (extracted ahead of time for the predicted pose) or stale (the map changed)
    bool fresh[3], rebuild[3] = {false, false, false};
    bool any = false;
    for (int k = 0; k < 3; ++k)
    {
      fresh[k] = UseKeypoints[k] && SpecBuilt[k];
      rebuild[k] = UseKeypoints[k] && !fresh[k] && !LocalMaps[k]->IsSubMapValid();
      any = any || fresh[k] || rebuild[k];
      SpecBuilt[k] = false;
    }
    float mn[9], mx[9];
    if (any && MapUpdate != MappingMode::NONE) LSA_TRY(lsa_working_bboxes(Ctx, Tworld.m, mn, mx));  // all types, one pass
    // A sub-map handed to the device ahead of time is only good for the prediction it was extracted for: whatever
    // does not hold gives it up here -- which also waits for the copy out of the staging buffer, so that the buffer
    // may be rewritten below.
    for (int k = 0; k < 3; ++k)
    {
      const bool holds = fresh[k] && MapUpdate != MappingMode::NONE && LocalMaps[k]->SubMapBuiltFor(mn + 3 * k, mx + 3 * k, KeypointCounts[k] / 2);
      if (SpecStaged[k] && !holds) LSA_TRY(lsa_drop_target_ahead(Ctx, LSA_TARGET_MAP, k));
      SpecStaged[k] = false;
      SpecDone[k].store(false);
    }
    for (int k = 0; k < 3; ++k)
    {
      if (!fresh[k] && !rebuild[k]) continue;
      RollingGrid* map = LocalMaps[k].get();
      const int minPts = KeypointCounts[k] / 2;
      if (fresh[k])
      {
        // the box under the actual pose touches the same voxels as the predicted one: the sub-map stands
        if (map->SubMapBuiltFor(mn + 3 * k, mx + 3 * k, minPts)) { Stats.submap_spec_hits++; SubMapSpecHitsTotal++; rebuild[k] = true; continue; }
        rebuild[k] = true;
      }

      if (MapUpdate == MappingMode::NONE)
        MapWorker[k].Submit([map] { map->BuildSubMap(); });
      else
      {
        const bool clear = map->IsTimeThreshold();
        const double now = CurrentTime;
        const float* lo3 = mn + 3 * k;
        const float* hi3 = mx + 3 * k;
        MapWorker[k].Submit([map, clear, now, minPts, mn0 = lo3[0], mn1 = lo3[1], mn2 = lo3[2], mx0 = hi3[0], mx1 = hi3[1], mx2 = hi3[2]] {
          if (clear) map->ClearOldPoints(now);
          const float lo[3] = {mn0, mn1, mn2}, hi[3] = {mx0, mx1, mx2};
          map->BuildSubMap(lo, hi, minPts);
        });
      }
    }
    for (int k = 0; k < 3; ++k)
    {
      if (!rebuild[k]) continue;
      MapWorker[k].Wait();
      // the map holds one point per leaf voxel: a search cell of about one leaf keeps a handful of candidates per cell
      lsa_set_target_cell_size(Ctx, LSA_TARGET_MAP, k, static_cast<float>((k == LSA_EDGE ? KnnCellScaleMapsEdges : KnnCellScaleMaps) * LocalMaps[k]->GetLeafSize()));
      // the worker extracted the sub-map straight into the target's pinned staging buffer: the copy is only enqueued
      LSA_TRY(lsa_set_target_staged(Ctx, LSA_TARGET_MAP, k, static_cast<int>(LocalMaps[k]->SubMapSize())));
    }
    Stats.submap += t.Stop();
  }

  TotalMatchedKeypoints = 0;
  lsa_match_params_t mp = LocMatchParams();
  Pose pendingD0 = Pose::Identity(), pendingD1 = Pose::Identity();
  bool pendingUndistort = false;  // the undistortion the last iteration ended with has not been applied yet
  bool postedUndistort = false;   // ... it was handed to the gate of the iteration enqueued ahead
  unsigned mask = 0;
  for (int k = 0; k < 3; ++k)
    if (UseKeypoints[k]) mask |= 1u << k;
  auto saturation = [&](unsigned icpIter) {
    const double iterRatio = icpIter / static_cast<double>(LocalizationICPMaxIter - 1);
    return (1 - iterRatio) * LocalizationInitSaturationDistance + iterRatio * LocalizationFinalSaturationDistance;
  };
  auto configure = [&](LocalOptimizer& optimizer) {
    optimizer.SetDeviceLoop(DeviceLM);
    optimizer.SetTwoDMode(TwoDMode);
    optimizer.SetLMMaxIter(LocalizationLMMaxIter);
    optimizer.SetMinMatches(MinNbMatchedKeypoints);
    optimizer.UseDeviceResiduals(7u);
  };
  // as in the ego-motion loop: iteration i + 1 behind a gate while iteration i runs.  The undistortion between two
  // iterations has to ride in the next search kernel for that (a launch of its own would have to be enqueued between
  // the two, when the motion is known).
  const bool undistortAhead = Undistortion == UNDISTORTION_REFINED;
  bool ahead = ICPAhead && DeviceLM && FusedMatch && (!undistortAhead || UndistortInSearch);
  static const int aheadLoops = std::getenv("LSA_ICP_AHEAD_LOOPS") ? std::atoi(std::getenv("LSA_ICP_AHEAD_LOOPS")) : 3;
  ahead = ahead && (aheadLoops & 2);
  // (ICPAhead = 2: the whole loop behind links, see ComputeEgoMotion; the device then also refines the undistortion)
  bool chain = ahead && ICPAhead >= 2 && LocalizationICPMaxIter <= kChainMax;
  if (chain) ahead = false;
  if (ahead || chain) lsa_icp_abandon(Ctx);
  unsigned chained = 0;
  long long chainSerial[kChainMax][3] = {};
  bool enqueued = false;
  long long aheadSerial[3] = {0, 0, 0};
  auto callOff = [&](int& ticket) {
    if (ticket < 0) return;
    lsa_icp_cancel(Ctx, ticket);
    lsa_solve_device_drop(Ctx);
    ticket = -1;
  };
  auto matchInLine = [&]() -> int {
    // the undistortion the previous iteration ended with rides in this iteration's search kernel (same keypoints, one launch less)
    if (pendingUndistort)
      LSA_TRY(lsa_match_types_undistorted(Ctx, LSA_TARGET_MAP, mask, &mp, Tworld.m, nullptr, pendingD0.m, pendingD1.m, Motion.Time0, Motion.Time1));
    else
      LSA_TRY(lsa_match_types(Ctx, LSA_TARGET_MAP, mask, LSA_SET_WORKING, &mp, Tworld.m, nullptr));
    pendingUndistort = false;
    for (int k = 0; k < 3; ++k)
      if ((mask >> k) & 1u) LocMatchSerial[k] = lsa_match_serial(Ctx, k);
    return LSA_OK;
  };

  for (unsigned icpIter = 0; icpIter < LocalizationICPMaxIter; ++icpIter)
  {
    Tick ticp;
    // the keyframe's insertion is handed to the maps' thread right after this loop: awake by then (last iteration: ~0.1 ms ahead)
    if (WorkerPrewake && DeviceMapsInUse() && icpIter + 1 == LocalizationICPMaxIter && MapUpdate != MappingMode::NONE) MapWorker[0].Expect(300e-6);
    mp.saturation_distance = saturation(icpIter);
    LocalOptimizer optimizer(Ctx);
    configure(optimizer);
    optimizer.SetPosePrior(Tworld);
    bool begun = enqueued;
    if (icpIter < chained)
    {
      begun = true;
      for (int k = 0; k < 3; ++k)
        if ((mask >> k) & 1u) LocMatchSerial[k] = chainSerial[icpIter][k];
    }
    else if (!enqueued)
    {
      LSA_TRY(matchInLine());
      if (chain)
      {
        double prior[6];
        ToXYZRPY(Tworld, prior);
        lsa_icp_link_t link;
        std::memset(&link, 0, sizeof(link));
        link.refine_undistortion = undistortAhead ? 1 : 0;
        link.have_log = LogTrajectory.empty() ? 0 : 1;
        link.prev_time = LogTrajectory.empty() ? 0. : LogTrajectory.back().time;
        link.cur_time = StampToSec(CurrentStamp);
        link.max_extrapolation_ratio = MaxExtrapolationRatio;
        std::memcpy(link.previous_world, PreviousTworld.m, sizeof(link.previous_world));
        {
          double* m = link.motion;
          m[0] = Motion.Time0; m[1] = Motion.Time1;
          m[2] = Motion.Rot0.w; m[3] = Motion.Rot0.x; m[4] = Motion.Rot0.y; m[5] = Motion.Rot0.z;
          m[6] = Motion.Rot1.w; m[7] = Motion.Rot1.x; m[8] = Motion.Rot1.y; m[9] = Motion.Rot1.z;
          for (int i = 0; i < 3; ++i) { m[10 + i] = Motion.Trans0[i]; m[13 + i] = Motion.Trans1[i]; }
        }
        for (unsigned j = icpIter; j < LocalizationICPMaxIter; ++j)
        {
          int leave = j + 1 < LocalizationICPMaxIter ? lsa_icp_link(Ctx) : -1;
          if (leave < 0) leave = -1;
          link.first = j == icpIter ? 1 : 0;
          const int rc = lsa_solve_device_begin_linked(Ctx, 7u, j == icpIter ? prior : nullptr, TwoDMode ? 1 : 0, static_cast<int>(LocalizationLMMaxIter), static_cast<int>(MinNbMatchedKeypoints), leave,
                                                       leave >= 0 ? &link : nullptr);
          if (rc < 0) { lsa_icp_abandon(Ctx); return Fail(rc, "lsa_solve_device_begin_linked"); }
          chained = j + 1;
          if (leave < 0) break;
          lsa_match_params_t next = mp;
          next.saturation_distance = saturation(j + 1);
          const int mrc = lsa_match_types_gated(Ctx, LSA_TARGET_MAP, mask, LSA_SET_WORKING, &next, undistortAhead ? 1 : 0);
          if (mrc < 0) { lsa_icp_abandon(Ctx); return Fail(mrc, "lsa_match_types_gated"); }
          if (mrc != 0) { lsa_icp_cancel(Ctx, leave); break; }
          for (int k = 0; k < 3; ++k)
            if ((mask >> k) & 1u) chainSerial[j + 1][k] = lsa_match_serial(Ctx, k);
        }
        chain = false;
        begun = true;
      }
      else if (ahead)
      {
        LSA_TRY(optimizer.Begin(false));
        begun = true;
      }
    }
    else
      for (int k = 0; k < 3; ++k)
        if ((mask >> k) & 1u) LocMatchSerial[k] = aheadSerial[k];
    enqueued = false;
    int ticket = -1;
    if (ahead && begun && icpIter + 1 < LocalizationICPMaxIter)
    {
      ticket = lsa_icp_gate(Ctx);
      if (ticket < 0) { ticket = -1; ahead = false; }
      else
      {
        lsa_match_params_t next = mp;
        next.saturation_distance = saturation(icpIter + 1);
        int rc = lsa_match_types_gated(Ctx, LSA_TARGET_MAP, mask, LSA_SET_WORKING, &next, undistortAhead ? 1 : 0);
        if (rc == 0)
        {
          for (int k = 0; k < 3; ++k)
            if ((mask >> k) & 1u) aheadSerial[k] = lsa_match_serial(Ctx, k);
          rc = lsa_solve_device_begin(Ctx, 7u, nullptr, TwoDMode ? 1 : 0, static_cast<int>(LocalizationLMMaxIter), static_cast<int>(MinNbMatchedKeypoints));
          if (rc < 0) { lsa_icp_cancel(Ctx, ticket); return Fail(rc, "lsa_solve_device_begin"); }
        }
        else
        {
          lsa_icp_cancel(Ctx, ticket);
          ticket = -1;
          ahead = false;
          if (rc < 0) return Fail(rc, "lsa_match_types_gated");
        }
      }
    }
    if (!begun) ArmLookaheadInterlude();
    Stats.loc_icp += ticp.Stop();
    Stats.loc_iters++;

    Tick tlm;
    SolveSummary summary;
    if (begun)
    {
      const int irc = InterludeWork();
      int rc = optimizer.End(summary);
      if (rc == LSA_E_GATE)
      {
        // (see the ego-motion loop) nothing of the iteration ran: done again in line, the rest of the loop without gates
        callOff(ticket);
        lsa_icp_abandon(Ctx);
        ahead = false;
        IcpGateTimeouts++;
        pendingUndistort = postedUndistort;  // (it was to ride in the search that did not run)
        rc = matchInLine();
        if (rc == LSA_OK) rc = optimizer.Solve(summary);
      }
      if (rc < 0) { callOff(ticket); lsa_icp_abandon(Ctx); return Fail(rc, "LocalOptimizer::Solve (localization)"); }
      if (rc == 1) { ticket = -1; ahead = false; chained = 0; }
      if (irc < 0) { callOff(ticket); return irc; }
    }
    else
    {
      LSA_TRY(optimizer.Solve(summary));
      LSA_TRY(FinishLookaheadInterlude());
    }
    TotalMatchedKeypoints = summary.num_matches;
    if (lsa_icp_trace_on())
      std::fprintf(stderr, "[icp] frame %u loc %u: matches %d evals %d steps %d cost %.17g -> %.17g%s\n", NbrFrameProcessed, icpIter, summary.num_matches, summary.num_evaluations,
                   summary.num_successful_steps, summary.initial_cost, summary.final_cost, summary.skipped ? " skipped" : "");
    if (summary.skipped)
    {
      // reset state to previous one to avoid instability (Slam.cxx:1098-1107)
      callOff(ticket);
      if (icpIter + 1 < chained) lsa_icp_abandon(Ctx);  // (the device has left go = 0 for them)
      Trelative = Pose::Identity();
      Tworld = PreviousTworld;
      if (Undistortion) Motion.SetTransforms(Pose::Identity(), Pose::Identity());
      LastError = "Not enough keypoints matched, Localization skipped for this frame.";
      Stats.loc_lm += tlm.Stop();
      break;
    }
    Stats.lm_evals += summary.num_evaluations;
    Tworld = optimizer.GetOptimizedPose();
    Trelative = Inverse(PreviousTworld) * Tworld;
    const bool lastIteration = (summary.num_successful_steps == 1) || (icpIter == LocalizationICPMaxIter - 1);
    if (lastIteration) callOff(ticket);  // nothing more to search: what follows on the stream does not wait behind the gate
    if (lastIteration && icpIter + 1 < chained) lsa_icp_abandon(Ctx);  // (links: the device has decided the same)
    if (Undistortion == UNDISTORTION_REFINED)
    {
      if (UndistortInSearch && !lastIteration)
      {
        RefineUndistortion(&pendingD0, &pendingD1);
        pendingUndistort = true;
      }
      else
      {
        int rc = RefineUndistortion();
        if (rc < 0) { callOff(ticket); return rc; }
      }
    }
    Stats.loc_lm += tlm.Stop();
    if (lastIteration)
    {
      Tick terr;
      LSA_TRY(optimizer.EstimateRegistrationError(LocalizationUncertainty));
      DbgAcc[4] += terr.Stop();
      break;
    }
    if (ticket >= 0)
    {
      double prior[6];
      ToXYZRPY(Tworld, prior);  // LocalOptimizer::SetPosePrior of the next iteration
      const int rc = lsa_icp_post(Ctx, ticket, Tworld.m, prior, pendingUndistort ? pendingD0.m : nullptr, pendingUndistort ? pendingD1.m : nullptr, Motion.Time0, Motion.Time1);
      if (rc < 0) return Fail(rc, "lsa_icp_post");
      postedUndistort = pendingUndistort;
      pendingUndistort = false;
      enqueued = true;
    }
    else if (icpIter + 1 < chained)
      pendingUndistort = false;  // the next search is in the queue and undistorts with what the device worked out (the same)
  }
  if (KeepMatchDebug)
    for (int k = 0; k < 3; ++k)
    {
      const int n = lsa_keypoint_count(Ctx, LSA_SET_WORKING, k);
      LocDebug[k].status.assign(n, LSA_MATCH_UNKOWN);
      LocDebug[k].weights.assign(n, 0.);
      if (n > 0) LSA_TRY(lsa_download_match(Ctx, k, LocDebug[k].status.data(), LocDebug[k].weights.data(), nullptr, n));
    }
  return LSA_OK;
}

int SlamCore::BeginSubMapSpeculation(const Pose& predicted)
{
  SpecPending = false;
  for (bool& b : DevSpec) b = false;
  if (MapUpdate == MappingMode::NONE) return LSA_OK;
  const bool onDevice = DeviceMapsInUse();
  if (onDevice)
  {
    // device maps: the sub-maps for the predicted boxes are extracted on the device, beside the ego-motion ICP; decaying
    // maps are left to the localization (ClearOldPoints comes first there)
    bool any = false;
    for (int k = 0; k < 3; ++k) any = any || (UseKeypoints[k] && KeypointCounts[k] > 0);
    if (!SubMapsAhead || !any || LocalMaps[0]->IsTimeThreshold()) return LSA_OK;
    // it has to be ready when the localization asks: where the ego-motion ICP is shorter than insertion + extraction +
    // grid build (small sensors) it is not, and is then left alone for a while
    if (DevSpecBackoff > 0) { --DevSpecBackoff; return LSA_OK; }
  }
  // Localization() will look at the keypoints after undistorting them with the motion between the previous pose
  // and the (then known) current one: the prediction does the same with the predicted pose -- the scan poses of
  // InterpolateScanPose at both ends of the keypoints' time range (Slam.cxx:1271-1285, 1288-1352).  Here Tworld
  // still is the previous frame's pose.
  bool interpolated = false;
  Pose begin = predicted, end = predicted;
  double t0 = 0., t1 = 0.;
  if (Undistortion && !LogTrajectory.empty())
  {
    LSA_TRY(lsa_keypoint_time_range(Ctx, LSA_SET_RAW_CURRENT, &t0, &t1));
    const double prevPoseTime = LogTrajectory.back().time;
    const double currPoseTime = StampToSec(CurrentStamp);
    if (t1 - t0 >= 1e-6 && currPoseTime != prevPoseTime)
    {
      auto scanPose = [&](double time) {
        if (std::abs(time / (currPoseTime - prevPoseTime)) > MaxExtrapolationRatio) return predicted;
        return LinearInterpolation(Tworld, predicted, currPoseTime + time, prevPoseTime, currPoseTime);
      };
      begin = scanPose(t0);
      end = scanPose(t1);
      interpolated = true;
    }
  }
  if (onDevice)
  {
    // Everything else is a dozen calls into the runtime: a host thread of its own
    // does it (the one that enqueues the next frame's ego-motion targets later in the frame), this one goes on to the
    // first search.  The boxes and the extractions go onto the grids' (look-ahead) stream, behind the previous keyframe's
    // insertions, the spare targets' search grids behind the extractions: nothing of it on the context's stream.
    int minPts[3];
    bool use[3];
    for (int k = 0; k < 3; ++k)
    {
      use[k] = UseKeypoints[k] && KeypointCounts[k] > 0;
      minPts[k] = KeypointCounts[k] / 2;
      if (use[k]) lsa_set_target_cell_size(Ctx, LSA_TARGET_MAP, k, static_cast<float>((k == LSA_EDGE ? KnnCellScaleMapsEdges : KnnCellScaleMaps) * LocalMaps[k]->GetLeafSize()));
      DevSpec[k] = use[k];
    }
    DevSpecStatus = 0;
    LSA_TRY(lsa_keypoint_boxes_predicted_mark(Ctx));  // the raw keypoints exist from here on the context's stream
    DevSpecRunning.store(true, std::memory_order_release);
    AheadWorker.Submit([this, interpolated, begin, end, t0, t1, use0 = use[0], use1 = use[1], use2 = use[2], m0 = minPts[0], m1 = minPts[1], m2 = minPts[2]] {
      const bool use[3] = {use0, use1, use2};
      const int minPts[3] = {m0, m1, m2};
      for (auto& w : MapWorker) w.Wait();  // the previous keyframe's insertions are on the grids' stream by now
      // the predicted boxes on the look-ahead stream, where the grids read them: nothing of this on the context's stream
      int rc = SpecBoxesOnLookahead ? lsa_keypoint_boxes_predicted(Ctx, LSA_SET_RAW_CURRENT, begin.m, interpolated ? end.m : nullptr, t0, t1)
                                    : (interpolated ? lsa_keypoint_bboxes_begin_interp(Ctx, LSA_SET_RAW_CURRENT, begin.m, end.m, t0, t1)
                                                    : lsa_keypoint_bboxes_begin(Ctx, LSA_SET_RAW_CURRENT, begin.m));
      for (int k = 0; k < 3 && rc >= 0; ++k)
        if (use[k]) rc = lsa_device_grid_submap_ahead_begin(DevMaps[k], k, minPts[k], k);
      // ... and waits for the extractions' sizes (this thread has nothing else to do) to enqueue the spare targets' search
      // grids at once.  The localization waits for this job to END (DevSpecRunning) before it touches the boxes' words or
      // the spare targets: nothing calls the job off half-way (the predicted boxes' kernels on the look-ahead stream have
      // to be over before lsa_keypoint_bboxes_begin rewrites those words on the context's stream)
      // (all the maps' grids in one sequence of launches, once the last size is there)
      lsa_device_grid* grids[3];
      int ng = 0;
      for (int k = 0; k < 3; ++k)
        if (use[k]) grids[ng++] = DevMaps[k];
      while (SpecGridsTogether && ng > 0 && rc >= 0)
      {
        rc = lsa_device_grid_submap_ahead_poll_all(grids, ng);
        if (rc != 1) break;
        std::this_thread::yield();
      }
      for (int i = 0; i < ng && !SpecGridsTogether && rc >= 0; ++i)
        while (rc >= 0)
        {
          rc = lsa_device_grid_submap_ahead_poll(grids[i]);
          if (rc != 1) break;
          std::this_thread::yield();
        }
      DevSpecStatus = rc < 0 ? rc : 0;
      DevSpecRunning.store(false, std::memory_order_release);
    });
    return LSA_OK;
  }
  if (interpolated) LSA_TRY(lsa_keypoint_bboxes_begin_interp(Ctx, LSA_SET_RAW_CURRENT, begin.m, end.m, t0, t1));
  else LSA_TRY(lsa_keypoint_bboxes_begin(Ctx, LSA_SET_RAW_CURRENT, predicted.m));
  SpecPending = true;
  return LSA_OK;
}

int SlamCore::FinishSubMapSpeculation()
{
  SpecPending = false;
  float mn[9], mx[9];
  LSA_TRY(lsa_keypoint_bboxes_end(Ctx, mn, mx));
  for (int k = 0; k < 3; ++k) { SpecDone[k].store(false); SpecStaged[k] = false; }
  for (int k = 0; k < 3; ++k)
  {
    if (!UseKeypoints[k] || KeypointCounts[k] <= 0) continue;
    RollingGrid* map = LocalMaps[k].get();
    const bool clear = map->IsTimeThreshold();
    const double now = CurrentTime;
    const int minPts = KeypointCounts[k] / 2;
    bool* built = &SpecBuilt[k];
    std::atomic<bool>* done = &SpecDone[k];
    const float* lo3 = mn + 3 * k;
    const float* hi3 = mx + 3 * k;
    // queued behind the previous keyframe's insertion on the same worker: it sees the final map
    MapWorker[k].Submit([map, clear, now, minPts, built, done, mn0 = lo3[0], mn1 = lo3[1], mn2 = lo3[2], mx0 = hi3[0], mx1 = hi3[1], mx2 = hi3[2]] {
      if (map->IsSubMapValid()) return;  // the map did not change: Slam.cxx:1013 keeps the kd-tree
      if (clear) map->ClearOldPoints(now);
      const float lo[3] = {mn0, mn1, mn2}, hi[3] = {mx0, mx1, mx2};
      map->BuildSubMap(lo, hi, minPts);
      *built = true;
      done->store(true, std::memory_order_release);
    });
  }
  return LSA_OK;
}

// Called between two steps of the ego-motion ICP: a sub-map the workers have finished for the predicted pose goes to
// the device right away -- upload and search grid on the look-ahead stream -- so that Localization(), if the
// prediction holds, only swaps it in.
int SlamCore::StageSpeculativeSubMaps()
{
  if (!BuildTargetsAhead) return LSA_OK;
  for (int k = 0; k < 3; ++k)
  {
    if (SpecStaged[k] || !SpecDone[k].load(std::memory_order_acquire)) continue;
    SpecStaged[k] = true;
    lsa_set_target_cell_size(Ctx, LSA_TARGET_MAP, k, static_cast<float>((k == LSA_EDGE ? KnnCellScaleMapsEdges : KnnCellScaleMaps) * LocalMaps[k]->GetLeafSize()));
    LSA_TRY(lsa_stage_target_ahead(Ctx, LSA_TARGET_MAP, k, static_cast<int>(LocalMaps[k]->SubMapSize())));
  }
  return LSA_OK;
}

// Slam::EstimateOverlap (Slam.cxx:1370-1388): Confidence::LCPEstimator on the registered frame
int SlamCore::EstimateOverlap()
{
  unsigned mask = 0;
  double leaf[3];
  for (int k = 0; k < 3; ++k)
  {
    leaf[k] = LocalMaps[k]->GetLeafSize();
    if (UseKeypoints[k] && (DeviceMapsInUse() ? lsa_target_size(Ctx, LSA_TARGET_MAP, k) > 0 : LocalMaps[k]->IsSubMapValid())) mask |= 1u << k;
  }
  if (!CurrentFrames.empty())
  {
    // several device frames: the estimator samples the aggregated registered cloud (Slam.cxx:1373); it is built
    // once (GetRegisteredFrame) and handed back as one frame of world points
    LSA_TRY(GetRegisteredFrame(Scratch));
    if (Scratch.empty()) { OverlapEstimation = -1.f; return LSA_OK; }
    LSA_TRY(lsa_frame_store_put(Ctx, kFirstDeviceFrameSlot + kMaxDeviceFrames, Scratch.data(), static_cast<int>(Scratch.size())));
    LSA_TRY(lsa_frame_store_use(Ctx, kFirstDeviceFrameSlot + kMaxDeviceFrames));
    const Pose identity = Pose::Identity();
    LSA_TRY(lsa_overlap(Ctx, mask, 0, identity.m, nullptr, 0., 0., OverlapSamplingRatio, leaf, &OverlapEstimation));
    return LSA_OK;
  }
  if (Undistortion)
  {
    const Pose H0 = (Tworld * Motion.GetH0()) * BaseToLidarOffset;
    const Pose H1 = (Tworld * Motion.GetH1()) * BaseToLidarOffset;
    LSA_TRY(lsa_overlap(Ctx, mask, 1, H0.m, H1.m, Motion.Time0, Motion.Time1, OverlapSamplingRatio, leaf, &OverlapEstimation));
  }
  else
  {
    const Pose tf = Tworld * BaseToLidarOffset;
    LSA_TRY(lsa_overlap(Ctx, mask, 0, tf.m, nullptr, 0., 0., OverlapSamplingRatio, leaf, &OverlapEstimation));
  }
  return LSA_OK;
}

// Slam::UpdateMapsUsingTworld (Slam.cxx:1178-1222)
int SlamCore::UpdateMapsUsingTworld()
{
  const Pose motion = Inverse(KfLastPose) * Tworld;
  const double trans = std::sqrt((motion(0, 3) * motion(0, 3) + motion(1, 3) * motion(1, 3)) + motion(2, 3) * motion(2, 3));
  const double rot = RotationAngle(motion);
  constexpr double MIN_KF_NB = 10.;
  WaitMaps();
  const double thresholdCoef = std::min(KfCounter / MIN_KF_NB, 1.);
  unsigned nbMapKpts = 0;
  const bool onDevice = DeviceMapsInUse();
  for (int k = 0; k < 3; ++k) nbMapKpts += onDevice ? static_cast<unsigned>(std::max(lsa_device_grid_size(DevMaps[k]), 0)) : LocalMaps[k]->Size();
  const bool isNewKeyFrame = nbMapKpts < MinNbMatchedKeypoints * 10 || trans >= thresholdCoef * KfDistanceThreshold ||
                             rot >= (thresholdCoef * KfAngleThreshold) / 180. * M_PI;
  if (!isNewKeyFrame) return LSA_OK;
  KfCounter++;
  KfLastPose = Tworld;
  for (double& v : MapJobSeconds) v = 0.;
  if (onDevice)
  {
    // the keyframe's keypoints go into the device maps as they are (WORLD transform, keying, sort, fold, merge: kernels on
    // the context's stream, nothing is waited for)
    // this thread only hands the keypoints over (one transform kernel per type); ONE worker enqueues the insertion of all
    // the types -- seven launches, a block row per map -- which runs on the look-ahead stream beside the next frame
    lsa_device_grid* grids[3];
    int types[3], ng = 0;
    for (int k = 0; k < 3; ++k)
    {
      if (!UseKeypoints[k]) continue;
      types[ng] = k;
      grids[ng++] = DevMaps[k];
    }
    if (ng > 0)
    {
      LSA_TRY(lsa_device_grid_stage_keypoints_all(grids, types, ng, LSA_SET_WORKING, Tworld.m));  // one transform launch for all the types
      const double time = CurrentTime;
      int* failed = &MapJobFailed[0];
      double* spent = &MapJobSeconds[0];
      MapWorker[0].Submit([g0 = grids[0], g1 = grids[1 % ng], g2 = grids[2 % ng], ng, time, failed, spent] {
        lsa_device_grid* gs[3] = {g0, g1, g2};
        Tick t;
        if (lsa_device_grid_add_staged_all(gs, ng, time) < 0) *failed = 1;
        *spent += t.Stop();  // host time of the enqueue (the kernels run on behind it)
      });
    }
    return LSA_OK;
  }
  // the device writes the world keypoints into pinned host memory; the map workers wait for exactly that and
  // insert them while this thread goes on with the next frame
  LSA_TRY(lsa_stage_transformed(Ctx, LSA_SET_WORKING, Tworld.m));
  for (int k = 0; k < 3; ++k)
  {
    if (!UseKeypoints[k]) continue;
    RollingGrid* map = LocalMaps[k].get();
    lsa_ctx* ctx = Ctx;
    const double time = CurrentTime;
    double* spent = &MapJobSeconds[k];
    MapWorker[k].Submit([map, ctx, k, time, spent] {
      map->WakeAddThreads();  // they are spinning by the time the staged keypoints have arrived
      const lsa_point_t* pts = nullptr;
      int n = 0;
      if (lsa_staged_transformed(ctx, k, &pts, &n) != LSA_OK) return;
      Tick t;
      map->Add(pts, static_cast<size_t>(std::max(n, 0)), false, time);
      *spent += t.Stop();
    });
  }
  return LSA_OK;
}

// Slam::LogCurrentFrameState (Slam.cxx:1225-1264); the keypoints log feeds the pose-graph optimisation, which is
// out of scope, so only the poses and their covariances are kept
void SlamCore::LogCurrentFrameState(double time)
{
  LogTrajectory.push_back({Tworld, time});
  if (LoggingTimeout != 0.)
  {
    LogCovariances.push_back(LocalizationUncertainty.Covariance);
    if (LoggingTimeout > 0)
      while (time - LogTrajectory.front().time > LoggingTimeout && LogTrajectory.size() > 2)
      {
        LogTrajectory.pop_front();
        LogCovariances.pop_front();
      }
  }
  else
    while (LogTrajectory.size() > 2) LogTrajectory.pop_front();
}

// Slam::CheckMotionLimits (Slam.cxx:1391-1484)
void SlamCore::CheckMotionLimits()
{
  const int nPoses = static_cast<int>(LogTrajectory.size());
  if (nPoses == 0) return;
  const double currentTimeStamp = StampToSec(CurrentStamp);
  double deltaTime = currentTimeStamp - LogTrajectory.back().time;
  double nextDeltaTime = std::numeric_limits<float>::max();
  int startIndex = nPoses - 1;  // the window ends on the current pose and starts at this logged one
  if (deltaTime < TimeWindowDuration)
  {
    // an interval [deltaTime, nextDeltaTime] containing TimeWindowDuration
    while (startIndex >= 0)
    {
      deltaTime = nextDeltaTime;
      nextDeltaTime = currentTimeStamp - LogTrajectory[startIndex].time;
      if (nextDeltaTime >= TimeWindowDuration) break;
      --startIndex;
    }
    if (startIndex < 0) startIndex = 0;  // not enough logged poses: the oldest one
    else if (std::abs(deltaTime - TimeWindowDuration) < std::abs(nextDeltaTime - TimeWindowDuration)) ++startIndex;
    deltaTime = currentTimeStamp - LogTrajectory[startIndex].time;
  }
  ComplyMotionLimits = true;
  const Pose window = Inverse(LogTrajectory[startIndex].pose) * Tworld;
  float angle = static_cast<float>(RotationAngle(window));  // [0, 2 pi]
  if (angle > M_PI) angle = static_cast<float>(2 * M_PI - angle);
  angle = static_cast<float>(angle / M_PI * 180.);  // Utils::Rad2Deg (Utilities.h:150-153)
  const float distance = static_cast<float>(std::sqrt(window.m[3] * window.m[3] + window.m[7] * window.m[7] + window.m[11] * window.m[11]));
  const float velocity[2] = {static_cast<float>(distance / deltaTime), static_cast<float>(angle / deltaTime)};
  if (NbrFrameProcessed >= 2)
  {
    bool comply = true;
    for (int i = 0; i < 2; ++i)
    {
      // Eigen::Array2f / double: the scalar is converted to the array's type first
      const float acceleration = (velocity[i] - PreviousVelocity[i]) / static_cast<float>(deltaTime);
      comply = comply && velocity[i] < VelocityLimits[i] && std::abs(acceleration) < AccelerationLimits[i];
    }
    ComplyMotionLimits = comply;
  }
  PreviousVelocity[0] = velocity[0];
  PreviousVelocity[1] = velocity[1];
  if (!ComplyMotionLimits) LastError = "The pose does not comply with the motion limitations. Lidar SLAM may have failed.";
}

// Slam::InterpolateScanPose (Slam.cxx:1271-1285)
Pose SlamCore::InterpolateScanPose(double time) const
{
  if (LogTrajectory.empty()) return Tworld;
  const double prevPoseTime = LogTrajectory.back().time;
  const double currPoseTime = StampToSec(CurrentStamp);
  if (std::abs(time / (currPoseTime - prevPoseTime)) > MaxExtrapolationRatio) return Tworld;
  return LinearInterpolation(PreviousTworld, Tworld, currPoseTime + time, prevPoseTime, currPoseTime);
}

// Slam::InitUndistortion (Slam.cxx:1288-1319)
int SlamCore::InitUndistortion()
{
  double t0, t1;
  LSA_TRY(lsa_working_time_range(Ctx, &t0, &t1));
  InitUndistortion(t0, t1);
  return LSA_OK;
}
void SlamCore::InitUndistortion(double t0, double t1)
{
  Motion.SetTimes(t0, t1);
  Motion.SetTransforms(Pose::Identity(), Pose::Identity());
  if (Motion.GetTimeRange() < 1e-6) Motion.SetTimes(0., 0.);
}

// Slam::RefineUndistortion (Slam.cxx:1322-1352); with d0 / d1 the two transforms are handed back instead of applied
int SlamCore::RefineUndistortion(Pose* outD0, Pose* outD1)
{
  const Pose previousBaseBegin = Motion.GetH0();
  const Pose previousBaseEnd = Motion.GetH1();
  const Pose worldToBaseBegin = InterpolateScanPose(Motion.Time0);
  const Pose worldToBaseEnd = InterpolateScanPose(Motion.Time1);
  const Pose baseToWorld = Inverse(Tworld);
  const Pose newBaseBegin = baseToWorld * worldToBaseBegin;
  const Pose newBaseEnd = baseToWorld * worldToBaseEnd;
  Motion.SetTransforms(newBaseBegin, newBaseEnd);
  const Pose d0 = newBaseBegin * Inverse(previousBaseBegin);
  const Pose d1 = newBaseEnd * Inverse(previousBaseEnd);
  if (outD0 && outD1) { *outD0 = d0; *outD1 = d1; return LSA_OK; }
  LSA_TRY(lsa_undistort(Ctx, d0.m, d1.m, Motion.Time0, Motion.Time1));
  return LSA_OK;
}

// Slam::GetMap / GetTargetSubMap (Slam.h:155-158)
int SlamCore::GetMap(int k, bool clean, std::vector<lsa_point_t>& out)
{
  if (!Ctx) return LSA_E_NO_DEVICE;
  if (DeviceMapsInUse())
  {
    WaitMaps();
    const int size = std::max(lsa_device_grid_size(DevMaps[k]), 0);
    out.resize(size);
    const int n = size > 0 ? lsa_device_grid_get(DevMaps[k], clean ? 1 : 0, out.data(), size) : 0;
    if (n < 0) return Fail(n, "lsa_device_grid_get");
    out.resize(n);
    return n;
  }
  out = Map(k).Get(clean);
  return static_cast<int>(out.size());
}

int SlamCore::GetTargetSubMap(int k, std::vector<lsa_point_t>& out)
{
  if (!Ctx) return LSA_E_NO_DEVICE;
  if (DeviceMapsInUse())
  {
    const int size = std::max(lsa_target_size(Ctx, LSA_TARGET_MAP, k), 0);
    out.resize(size);
    if (size > 0) LSA_TRY(lsa_download_target(Ctx, LSA_TARGET_MAP, k, out.data(), size));
    return size;
  }
  const RollingGrid& map = Map(k);
  out.assign(map.SubMapData(), map.SubMapData() + map.SubMapSize());
  return static_cast<int>(out.size());
}

int SlamCore::GetKeypoints(int type, bool world, std::vector<lsa_point_t>& out)
{
  if (!Ctx) return LSA_E_NO_DEVICE;
  const int n = lsa_keypoint_count(Ctx, LSA_SET_WORKING, type);
  out.resize(std::max(n, 0));
  if (n <= 0) return 0;
  if (world) LSA_TRY(lsa_download_transformed(Ctx, LSA_SET_WORKING, type, Tworld.m, out.data(), n));
  else LSA_TRY(lsa_download_keypoints(Ctx, LSA_SET_WORKING, type, out.data(), n));
  return n;
}

int SlamCore::GetRawKeypoints(int type, std::vector<lsa_point_t>& out)
{
  if (!Ctx) return LSA_E_NO_DEVICE;
  const int n = lsa_keypoint_count(Ctx, LSA_SET_RAW_CURRENT, type);
  out.resize(std::max(n, 0));
  if (n > 0) LSA_TRY(lsa_download_keypoints(Ctx, LSA_SET_RAW_CURRENT, type, out.data(), n));
  return n;
}

// Slam::GetRegisteredFrame -> AggregateFrames(CurrentFrames, true) (Slam.cxx:660-667, 1512-1578)
int SlamCore::GetRegisteredFrame(std::vector<lsa_point_t>& out)
{
  if (!Ctx) return LSA_E_NO_DEVICE;
  out.clear();
  if (!HaveFrame) return 0;
  if (!CurrentFrames.empty())
  {
    // AggregateFrames(CurrentFrames, true): every device frame with its own offset and time shift (Slam.cxx:1512-1578)
    size_t total = 0;
    for (const HeldFrame& f : CurrentFrames) total += static_cast<size_t>(f.n);
    out.resize(total);
    size_t at = 0;
    for (const HeldFrame& f : CurrentFrames)
    {
      LSA_TRY(lsa_frame_store_use(Ctx, f.slot));
      const Pose offset = GetBaseToLidarOffset(f.device);
      if (Undistortion)
      {
        const Pose H0 = (Tworld * Motion.GetH0()) * offset;
        const Pose H1 = (Tworld * Motion.GetH1()) * offset;
        LSA_TRY(lsa_transform_frame_at(Ctx, 1, H0.m, H1.m, Motion.Time0, Motion.Time1, f.timeOffset, out.data() + at, f.n));
      }
      else
      {
        const Pose tf = Tworld * offset;
        LSA_TRY(lsa_transform_frame_at(Ctx, 0, tf.m, nullptr, 0., 0., f.timeOffset, out.data() + at, f.n));
      }
      at += static_cast<size_t>(f.n);
    }
    return static_cast<int>(total);
  }
  const int n = lsa_frame_size(Ctx);
  if (n <= 0) return 0;
  out.resize(n);
  if (Undistortion)
  {
    const Pose H0 = (Tworld * Motion.GetH0()) * BaseToLidarOffset;
    const Pose H1 = (Tworld * Motion.GetH1()) * BaseToLidarOffset;
    LSA_TRY(lsa_transform_frame(Ctx, 1, H0.m, H1.m, Motion.Time0, Motion.Time1, out.data(), n));
  }
  else
  {
    // the reference skips the work when the transform is identity; applying identity gives the same points
    const Pose tf = Tworld * BaseToLidarOffset;
    LSA_TRY(lsa_transform_frame(Ctx, 0, tf.m, nullptr, 0., 0., out.data(), n));
  }
  return n;
}

// name -> member table shared by SetParam / GetParam
#define LSA_PARAMS(X)                                                                                   \
  X("UseBlobs", UseKeypoints[LSA_BLOB], bool)                                                          \
  X("UseEdges", UseKeypoints[LSA_EDGE], bool)                                                          \
  X("UsePlanes", UseKeypoints[LSA_PLANE], bool)                                                        \
  X("TwoDMode", TwoDMode, bool)                                                                        \
  X("BuildTargetsAhead", BuildTargetsAhead, bool)                                                      \
  X("DeviceLM", DeviceLM, bool)                                                                        \
  X("MapsOnDevice", MapsOnDevice, bool)                                                                \
  X("SubMapsAhead", SubMapsAhead, bool)                                                                \
  X("SubMapsAheadAdaptive", SubMapsAheadAdaptive, bool)                                                \
  X("LocalizationStartFused", LocalizationStartFused, bool)                                            \
  X("WorkerPrewake", WorkerPrewake, bool)                                                              \
  X("ICPAhead", ICPAhead, int)                                                                         \
  X("UndistortInSearch", UndistortInSearch, bool)                                                      \
  X("SpecBoxesOnLookahead", SpecBoxesOnLookahead, bool)                                                \
  X("SpecGridsTogether", SpecGridsTogether, bool)                                                      \
  X("FusedMatch", FusedMatch, bool)                                                                    \
  X("EgoMotionICPMaxIter", EgoMotionICPMaxIter, unsigned)                                              \
  X("LocalizationICPMaxIter", LocalizationICPMaxIter, unsigned)                                        \
  X("EgoMotionLMMaxIter", EgoMotionLMMaxIter, unsigned)                                                \
  X("LocalizationLMMaxIter", LocalizationLMMaxIter, unsigned)                                          \
  X("EgoMotionMaxNeighborsDistance", EgoMotionMaxNeighborsDistance, double)                            \
  X("LocalizationMaxNeighborsDistance", LocalizationMaxNeighborsDistance, double)                      \
  X("EgoMotionEdgeNbNeighbors", EgoMotionEdgeNbNeighbors, unsigned)                                    \
  X("EgoMotionEdgeMinNbNeighbors", EgoMotionEdgeMinNbNeighbors, unsigned)                              \
  X("EgoMotionEdgeMaxModelError", EgoMotionEdgeMaxModelError, double)                                  \
  X("EgoMotionPlaneNbNeighbors", EgoMotionPlaneNbNeighbors, unsigned)                                  \
  X("EgoMotionPlanarityThreshold", EgoMotionPlanarityThreshold, double)                                \
  X("EgoMotionPlaneMaxModelError", EgoMotionPlaneMaxModelError, double)                                \
  X("EgoMotionInitSaturationDistance", EgoMotionInitSaturationDistance, double)                        \
  X("EgoMotionFinalSaturationDistance", EgoMotionFinalSaturationDistance, double)                      \
  X("LocalizationEdgeNbNeighbors", LocalizationEdgeNbNeighbors, unsigned)                              \
  X("LocalizationEdgeMinNbNeighbors", LocalizationEdgeMinNbNeighbors, unsigned)                        \
  X("LocalizationEdgeMaxModelError", LocalizationEdgeMaxModelError, double)                            \
  X("LocalizationPlaneNbNeighbors", LocalizationPlaneNbNeighbors, unsigned)                            \
  X("LocalizationPlanarityThreshold", LocalizationPlanarityThreshold, double)                          \
  X("LocalizationPlaneMaxModelError", LocalizationPlaneMaxModelError, double)                          \
  X("LocalizationBlobNbNeighbors", LocalizationBlobNbNeighbors, unsigned)                              \
  X("LocalizationInitSaturationDistance", LocalizationInitSaturationDistance, double)                  \
  X("LocalizationFinalSaturationDistance", LocalizationFinalSaturationDistance, double)                \
  X("MaxExtrapolationRatio", MaxExtrapolationRatio, double)                                            \
  X("MinNbMatchedKeypoints", MinNbMatchedKeypoints, unsigned)                                          \
  X("KfDistanceThreshold", KfDistanceThreshold, double)                                                \
  X("KfAngleThreshold", KfAngleThreshold, double)                                                      \
  X("KeepMatchDebug", KeepMatchDebug, bool)                                                            \
  X("KnnCellSizeEgoMotion", KnnCellSizeEgoMotion, double)                                              \
  X("KnnCellScaleMaps", KnnCellScaleMaps, double)                                                      \
  X("KnnCellSizeEgoMotionEdges", KnnCellSizeEgoMotionEdges, double)                                    \
  X("KnnCellScaleMapsEdges", KnnCellScaleMapsEdges, double)                                            \
  X("KnnLanesEdges", KnnLanesEdges, int)                                                               \
  X("KnnLanesPlanes", KnnLanesPlanes, int)                                                             \
  X("KnnLanesBlobs", KnnLanesBlobs, int)                                                               \
  X("KnnRoundsEdges", KnnRoundsEdges, int)                                                             \
  X("KnnRoundsPlanes", KnnRoundsPlanes, int)                                                           \
  X("KnnRoundsBlobs", KnnRoundsBlobs, int)                                                             \
  X("NeighborWidth", ExtractParams.neighbor_width, int)                                                \
  X("MinDistanceToSensor", ExtractParams.min_distance_to_sensor, float)                                \
  X("MinBeamSurfaceAngle", ExtractParams.min_beam_surface_angle, float)                                \
  X("PlaneSinAngleThreshold", ExtractParams.plane_sin_angle_threshold, float)                          \
  X("EdgeSinAngleThreshold", ExtractParams.edge_sin_angle_threshold, float)                            \
  X("DistToLineThreshold", ExtractParams.dist_to_line_threshold, float)                                \
  X("EdgeDepthGapThreshold", ExtractParams.edge_depth_gap_threshold, float)                            \
  X("EdgeSaliencyThreshold", ExtractParams.edge_saliency_threshold, float)                             \
  X("EdgeIntensityGapThreshold", ExtractParams.edge_intensity_gap_threshold, float)

// The maps live either in the device grids or in the host grids ("MapsOnDevice" = 0): a setter that
// moves them from one side to the other takes the points along, the way RollingGrid's own geometry setters do
// (prevMap = Get(); Clear(); Add(prevMap), RollingGrid.cxx:59-88: counts start again, the points keep their place).
int SlamCore::MigrateMaps(bool fromDevice)
{
  for (int k = 0; k < 3; ++k)
  {
    if (!DevMaps[k]) continue;
    if (fromDevice)
    {
      const int size = std::max(lsa_device_grid_size(DevMaps[k]), 0);
      std::vector<lsa_point_t> pts(static_cast<size_t>(size));
      const int n = size > 0 ? lsa_device_grid_get(DevMaps[k], 0, pts.data(), size) : 0;
      if (n < 0) return Fail(n, "lsa_device_grid_get");
      LocalMaps[k]->Clear();
      LocalMaps[k]->Add(pts.data(), static_cast<size_t>(n), false, CurrentTime);
      LSA_TRY(lsa_device_grid_clear(DevMaps[k]));
    }
    else
    {
      const RollingGrid::PointCloud pts = LocalMaps[k]->Get();
      LSA_TRY(lsa_device_grid_clear(DevMaps[k]));
      if (!pts.empty()) LSA_TRY(lsa_device_grid_add(DevMaps[k], pts.data(), static_cast<int>(pts.size()), 0, CurrentTime, 1));
      LocalMaps[k]->Clear();
    }
  }
  return LSA_OK;
}

int SlamCore::SetParam(const std::string& name, double v)
{
  WaitMaps();
  const bool onDevice = DeviceMapsInUse();
  const int rc = SetParamValue(name, v);
  if (rc == LSA_OK && DeviceMapsInUse() != onDevice) return MigrateMaps(onDevice);
  return rc;
}

int SlamCore::SetParamValue(const std::string& name, double v)
{
#define X(NAME, MEMBER, TYPE) if (name == NAME) { MEMBER = static_cast<TYPE>(v); return LSA_OK; }
  LSA_PARAMS(X)
#undef X
  if (name == "NbThreads") return LSA_OK;  // OpenMP thread count of the reference: no meaning on the device path
  if (name == "Verbosity") return LSA_OK;
  if (name == "EgoMotion") { EgoMotion = static_cast<EgoMotionMode>(static_cast<int>(v)); return LSA_OK; }
  if (name == "Undistortion") { Undistortion = static_cast<UndistortionMode>(static_cast<int>(v)); return LSA_OK; }
  if (name == "MapUpdate") { MapUpdate = static_cast<MappingMode>(static_cast<int>(v)); return LSA_OK; }
  if (name == "OverlapSamplingRatio")
  {
    // Slam::SetOverlapSamplingRatio (Slam.cxx:1359-1367)
    OverlapSamplingRatio = std::min(std::max(static_cast<float>(v), 0.f), 1.f);
    if (OverlapSamplingRatio == 0.f) OverlapEstimation = -1.f;
    return LSA_OK;
  }
  if (name == "MapAddThreads") { LocalMaps[LSA_PLANE]->SetAddThreads(static_cast<int>(v)); return LSA_OK; }
  if (name == "MapAddThreadsEdges") { LocalMaps[LSA_EDGE]->SetAddThreads(static_cast<int>(v)); return LSA_OK; }
  if (name == "LoggingTimeout") { LoggingTimeout = v; return LSA_OK; }
  if (name == "TimeWindowDuration") { TimeWindowDuration = static_cast<float>(v); return LSA_OK; }
  if (name == "VelocityLimitLinear") { VelocityLimits[0] = static_cast<float>(v); return LSA_OK; }
  if (name == "VelocityLimitAngular") { VelocityLimits[1] = static_cast<float>(v); return LSA_OK; }
  if (name == "AccelerationLimitLinear") { AccelerationLimits[0] = static_cast<float>(v); return LSA_OK; }
  if (name == "AccelerationLimitAngular") { AccelerationLimits[1] = static_cast<float>(v); return LSA_OK; }
  if (name == "AzimuthalResolution") { if (Ctx) lsa_set_azimuthal_resolution(Ctx, static_cast<float>(v)); return LSA_OK; }
  auto dev = [&](int k, const char* what, double value) { if (DevMaps[k]) lsa_device_grid_set(DevMaps[k], what, value); };
  if (name == "VoxelGridLeafSizeEdges") { LocalMaps[LSA_EDGE]->SetLeafSize(v); dev(LSA_EDGE, "LeafSize", v); return LSA_OK; }
  if (name == "VoxelGridLeafSizePlanes") { LocalMaps[LSA_PLANE]->SetLeafSize(v); dev(LSA_PLANE, "LeafSize", v); return LSA_OK; }
  if (name == "VoxelGridLeafSizeBlobs") { LocalMaps[LSA_BLOB]->SetLeafSize(v); dev(LSA_BLOB, "LeafSize", v); return LSA_OK; }
  if (name == "VoxelGridSize") { for (int k = 0; k < 3; ++k) { LocalMaps[k]->SetGridSize(static_cast<int>(v)); dev(k, "GridSize", v); } return LSA_OK; }
  if (name == "VoxelGridResolution") { for (int k = 0; k < 3; ++k) { LocalMaps[k]->SetVoxelResolution(v); dev(k, "VoxelResolution", v); } return LSA_OK; }
  if (name == "VoxelGridMinFramesPerVoxel") { for (int k = 0; k < 3; ++k) { LocalMaps[k]->SetMinFramesPerVoxel(static_cast<unsigned>(v)); dev(k, "MinFramesPerVoxel", v); } return LSA_OK; }
  if (name == "VoxelGridDecayingThreshold") { for (int k = 0; k < 3; ++k) { LocalMaps[k]->SetDecayingThreshold(v); dev(k, "DecayingThreshold", v); } return LSA_OK; }
  if (name == "VoxelGridSamplingMode") { for (int k = 0; k < 3; ++k) { LocalMaps[k]->SetSampling(static_cast<SamplingMode>(static_cast<int>(v))); dev(k, "Sampling", v); } return LSA_OK; }
  if (name == "OrderedMaps") { OrderedMaps = v != 0; for (auto& m : LocalMaps) m->SetOrdered(OrderedMaps); return LSA_OK; }
  LastError = "unknown parameter " + name;
  return LSA_E_ARG;
}

int SlamCore::GetParam(const std::string& name, double* v) const
{
  if (!v) return LSA_E_ARG;
#define X(NAME, MEMBER, TYPE) if (name == NAME) { *v = static_cast<double>(MEMBER); return LSA_OK; }
  LSA_PARAMS(X)
#undef X
  if (name == "EgoMotion") { *v = static_cast<int>(EgoMotion); return LSA_OK; }
  if (name == "Undistortion") { *v = static_cast<int>(Undistortion); return LSA_OK; }
  if (name == "MapUpdate") { *v = static_cast<int>(MapUpdate); return LSA_OK; }
  if (name == "AzimuthalResolution") { *v = Ctx ? lsa_get_azimuthal_resolution(Ctx) : 0.; return LSA_OK; }
  if (name == "VoxelGridLeafSizeEdges") { *v = LocalMaps[LSA_EDGE]->GetLeafSize(); return LSA_OK; }
  if (name == "VoxelGridLeafSizePlanes") { *v = LocalMaps[LSA_PLANE]->GetLeafSize(); return LSA_OK; }
  if (name == "VoxelGridLeafSizeBlobs") { *v = LocalMaps[LSA_BLOB]->GetLeafSize(); return LSA_OK; }
  if (name == "VoxelGridSize") { *v = LocalMaps[0]->GetGridSize(); return LSA_OK; }
  if (name == "VoxelGridResolution") { *v = LocalMaps[0]->GetVoxelResolution(); return LSA_OK; }
  if (name == "VoxelGridSamplingMode") { *v = static_cast<int>(LocalMaps[0]->GetSampling()); return LSA_OK; }
  if (name == "VoxelGridDecayingThreshold") { *v = LocalMaps[0]->GetDecayingThreshold(); return LSA_OK; }
  if (name == "VoxelGridMinFramesPerVoxel") { *v = LocalMaps[0]->GetMinFramesPerVoxel(); return LSA_OK; }
  if (name == "NbrFrameProcessed") { *v = NbrFrameProcessed; return LSA_OK; }
  if (name == "TotalMatchedKeypoints") { *v = TotalMatchedKeypoints; return LSA_OK; }
  if (name == "OverlapSamplingRatio") { *v = OverlapSamplingRatio; return LSA_OK; }
  if (name == "OverlapEstimation") { *v = OverlapEstimation; return LSA_OK; }
  if (name == "SubMapSpeculationHits") { *v = SubMapSpecHitsTotal; return LSA_OK; }
  if (name == "LookaheadAdopted") { *v = Ctx ? lsa_extract_prefetch_adopted(Ctx) : 0; return LSA_OK; }
  if (name == "OrderedMaps") { *v = OrderedMaps ? 1. : 0.; return LSA_OK; }
  if (name == "DeviceMapsInUse") { *v = DeviceMapsInUse() ? 1. : 0.; return LSA_OK; }
  if (name == "UploadsAdopted") { *v = Ctx ? lsa_uploads_adopted(Ctx) : 0; return LSA_OK; }
  if (name == "DeviceSolveFallbacks") { *v = Ctx ? lsa_solve_device_fallbacks(Ctx) : 0; return LSA_OK; }
  if (name == "IcpGateTimeouts") { *v = IcpGateTimeouts; return LSA_OK; }
  if (name == "TargetsBuiltAheadAdopted") { *v = Ctx ? lsa_prepared_targets_adopted(Ctx) : 0; return LSA_OK; }
  if (name == "SubMapsStagedAheadAdopted") { *v = Ctx ? lsa_staged_targets_adopted(Ctx) : 0; return LSA_OK; }
  if (name == "MapAddThreads") { *v = LocalMaps[LSA_PLANE]->GetAddThreads(); return LSA_OK; }
  if (name == "MapAddThreadsEdges") { *v = LocalMaps[LSA_EDGE]->GetAddThreads(); return LSA_OK; }
  if (name == "LoggingTimeout") { *v = LoggingTimeout; return LSA_OK; }
  if (name == "TimeWindowDuration") { *v = TimeWindowDuration; return LSA_OK; }
  if (name == "VelocityLimitLinear") { *v = VelocityLimits[0]; return LSA_OK; }
  if (name == "VelocityLimitAngular") { *v = VelocityLimits[1]; return LSA_OK; }
  if (name == "AccelerationLimitLinear") { *v = AccelerationLimits[0]; return LSA_OK; }
  if (name == "AccelerationLimitAngular") { *v = AccelerationLimits[1]; return LSA_OK; }
  if (name == "ComplyMotionLimits") { *v = ComplyMotionLimits ? 1. : 0.; return LSA_OK; }
  if (name == "Latency") { *v = Latency; return LSA_OK; }
  return LSA_E_ARG;
}

}  // namespace host
}  // namespace lsa
