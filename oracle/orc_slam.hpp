// ORACLE -- TEST INFRASTRUCTURE ONLY (see orc_math.hpp header).  PARITY UNPINNED.
//
// CPU restatement of the per-frame pipeline of LidarSlam::Slam
//   slam_lib/src/Slam.cxx:143-210 (ctor/Reset), 230-344 (AddFrames), 709-810
//   (CheckFrames, ExtractKeypoints), 813-972 (ComputeEgoMotion), 975-1175
//   (Localization), 1178-1264 (maps update, logging), 1271-1352 (undistortion),
//   1491-1578 (transform helpers); defaults slam_lib/include/LidarSlam/Slam.h:403-694.
// One LiDAR device per call (AddFrame); sensor constraints, pose graph, PCD IO,
// overlap estimator and motion limits are outside the hot path.
#pragma once
#include <array>
#include <cfloat>
#include <deque>
#include <map>
#include <memory>
#include <string>
#include "orc_math.hpp"
#include "orc_extractor.hpp"
#include "orc_matcher.hpp"
#include "orc_lm.hpp"
#include "orc_rolling_grid.hpp"

namespace orc
{

enum UndistortionMode { UNDIST_NONE = 0, UNDIST_ONCE = 1, UNDIST_REFINED = 2 };
enum class EgoMotionMode { NONE = 0, MOTION_EXTRAPOLATION = 1, REGISTRATION = 2, MOTION_EXTRAPOLATION_AND_REGISTRATION = 3 };
enum class MappingMode { NONE = 0, ADD_KPTS_TO_FIXED_MAP = 1, UPDATE = 2 };

struct StampedPose { Iso pose; double time; };

// wall-clock seconds per stage, placed like the reference's Utils::Timer calls
struct StageTimes
{
  double total = 0, extract = 0, ego_icp = 0, ego_lm = 0, loc_icp = 0, loc_lm = 0, undistort = 0, submap = 0, maps = 0;
  int ego_iters = 0, loc_iters = 0, lm_evals = 0;
};

class Slam
{
public:
  Slam();
  void Reset(bool resetLog = true);
  void AddFrame(const std::vector<Point>& frame, uint64_t stampUs, unsigned seq = 0);
  // Slam::AddFrames: one frame per LiDAR device, each with its own header stamp (Slam.cxx:230-344)
  void AddFrames(const std::vector<const std::vector<Point>*>& frames, const std::vector<uint64_t>& stampsUs);

  Iso GetWorldTransform() const { return LogTrajectory.empty() ? iso_identity() : LogTrajectory.back().pose; }
  Iso GetLatencyCompensatedWorldTransform() const;          // Slam.cxx:555-590
  void SetWorldTransformFromGuess(const Iso& guess);          // Slam.cxx:490-501
  void GetDebugInformation(double out[10]) const;             // Slam.cxx:610-633
  const double* GetTransformCovariance() const { return LocalizationUncertainty.Covariance; }

  // parameters (Slam.h:403-694)
  int NbThreads = 1;
  bool UseKeypoints[3] = {true, true, false};
  EgoMotionMode EgoMotion = EgoMotionMode::MOTION_EXTRAPOLATION;
  UndistortionMode Undistortion = UNDIST_REFINED;
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
  Iso BaseToLidarOffset = iso_identity();  // device 0
  std::map<int, Iso> OtherBaseToLidarOffsets;
  Iso GetBaseToLidarOffset(int deviceId) const;  // identity for a device nobody configured

  Extractor KeyPointsExtractor;                // device 0 (Slam::Slam allocates it, Slam.cxx:146)
  std::map<int, Extractor> OtherExtractors;    // SetKeyPointsExtractor(extractor, deviceId != 0)
  std::shared_ptr<RollingGrid> LocalMaps[3];

  // state readable by the tests
  Iso Tworld, PreviousTworld, Trelative;
  std::vector<Point> CurrentRawKeypoints[3], PreviousRawKeypoints[3], CurrentUndistortedKeypoints[3], CurrentWorldKeypoints[3];
  MatchingResults EgoMotionMatchingResults[2], LocalizationMatchingResults[3];
  RegistrationError LocalizationUncertainty;
  unsigned TotalMatchedKeypoints = 0;
  unsigned NbrFrameProcessed = 0;
  int KfCounter = 0;
  StageTimes Times;
  std::deque<StampedPose> LogTrajectory;

  std::vector<Point> GetRegisteredFrame();

  // Confidence estimator (Slam.h OverlapSamplingRatio / GetOverlapEstimation; Slam.cxx:1359-1388)
  float OverlapSamplingRatio = 0.f;
  float OverlapEstimation = -1.f;
  // Slam.h:663-694, Slam.cxx:1391-1484
  float VelocityLimits[2] = {FLT_MAX, FLT_MAX}, AccelerationLimits[2] = {FLT_MAX, FLT_MAX};
  float TimeWindowDuration = 0.f;
  bool ComplyMotionLimits = true;
  float PreviousVelocity[2] = {0.f, 0.f};
  double LoggingTimeout = 0.;  // Slam.h:425-438
  double Latency = 0.;         // the tests set it: the reference measures it
  std::deque<std::array<double, 36>> LogCovariances;

private:
  void CheckMotionLimits();
  void EstimateOverlap();
  bool CheckFrame(const std::vector<Point>& frame, uint64_t stampUs);
  void ExtractKeypoints();
  void ComputeEgoMotion();
  void Localization();
  void UpdateMapsUsingTworld();
  void LogCurrentFrameState(double time);
  Iso InterpolateScanPose(double time);
  void InitUndistortion();
  void RefineUndistortion();
  std::vector<Point> AggregateKeypoints(const std::vector<Point>& kpts) const;
  // AggregateFrames (Slam.cxx:1512-1578) over clouds that carry a header stamp each
  struct StampedCloud
  {
    const std::vector<Point>* cloud;
    uint64_t stampUs;
  };
  std::vector<Point> AggregateFrames(const std::vector<StampedCloud>& frames, bool worldCoordinates) const;
  std::vector<StampedCloud> CurrentFrames;  // all the frames of the current AddFrames call

  const std::vector<Point>* CurrentFrame = nullptr;
  uint64_t CurrentStamp = 0;
  bool HasFrame = false;
  double CurrentTime = 0.;
  Interpolator WithinFrameMotion;
  Iso KfLastPose;
  KDTree EgoTrees[2];
};

}  // namespace orc
