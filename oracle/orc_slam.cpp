// ORACLE -- TEST INFRASTRUCTURE ONLY (see orc_math.hpp header).  PARITY UNPINNED.
//
// CPU restatement of slam_lib/src/Slam.cxx (hot-path subset, see orc_slam.hpp).
#include "orc_slam.hpp"
#include <chrono>
#include <cstdio>

namespace orc
{
namespace
{
struct Tick
{
  std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
  double Stop() const { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); }
};
inline double StampToSec(uint64_t us) { return us * 1e-6; }
}  // namespace

// Slam.cxx:143-161
Slam::Slam()
{
  for (int k = 0; k < 3; ++k) LocalMaps[k] = std::make_shared<RollingGrid>();
  for (int k = 0; k < 3; ++k) LocalMaps[k]->SetVoxelResolution(10.);
  for (int k = 0; k < 3; ++k) LocalMaps[k]->SetGridSize(50);
  LocalMaps[EDGE]->SetLeafSize(0.30);
  LocalMaps[PLANE]->SetLeafSize(0.60);
  LocalMaps[BLOB]->SetLeafSize(0.30);
  Reset();
}

// Slam.cxx:164-210
void Slam::Reset(bool resetLog)
{
  for (int k = 0; k < 3; ++k) LocalMaps[k]->Reset();
  KfLastPose = iso_identity();
  KfCounter = 0;
  Tworld = PreviousTworld = Trelative = iso_identity();
  WithinFrameMotion.SetTransforms(iso_identity(), iso_identity());
  LocalizationUncertainty = RegistrationError();
  CurrentFrame = nullptr;
  CurrentStamp = 0;  // the "previous" frame after a reset is an empty cloud with stamp 0
  for (int k = 0; k < 3; ++k)
  {
    CurrentRawKeypoints[k].clear();
    CurrentUndistortedKeypoints[k].clear();
    CurrentWorldKeypoints[k].clear();
  }
  for (int k = 0; k < 2; ++k) EgoMotionMatchingResults[k] = MatchingResults();
  for (int k = 0; k < 3; ++k) LocalizationMatchingResults[k] = MatchingResults();
  if (resetLog)
  {
    NbrFrameProcessed = 0;
    LogTrajectory.clear();
  }
}

// Slam.cxx:709-743
bool Slam::CheckFrame(const std::vector<Point>& frame, uint64_t stampUs)
{
  if (frame.empty())
    return false;
  if (stampUs == CurrentStamp)
    return false;
  return true;
}

// Slam.cxx:230-344
void Slam::AddFrame(const std::vector<Point>& frame, uint64_t stampUs, unsigned)
{
  Tick total;
  Times = StageTimes();
  if (!CheckFrame(frame, stampUs))
    return;
  CurrentFrame = &frame;
  CurrentFrames.clear();
  CurrentStamp = stampUs;
  CurrentTime = StampToSec(stampUs);

  { Tick t; ExtractKeypoints(); Times.extract = t.Stop(); }
  ComputeEgoMotion();
  Localization();
  // confidence estimators: before the maps update, which resets the kd-trees (Slam.cxx:269-280)
  if (OverlapSamplingRatio > 0) EstimateOverlap();
  if (TimeWindowDuration > 0) CheckMotionLimits();
  if (MapUpdate == MappingMode::ADD_KPTS_TO_FIXED_MAP || MapUpdate == MappingMode::UPDATE)
  {
    Tick t;
    UpdateMapsUsingTworld();
    Times.maps = t.Stop();
  }
  LogCurrentFrameState(CurrentTime);
  NbrFrameProcessed++;
  Times.total = total.Stop();
}

Iso Slam::GetBaseToLidarOffset(int deviceId) const
{
  if (deviceId == 0) return BaseToLidarOffset;
  auto it = OtherBaseToLidarOffsets.find(deviceId);
  return it == OtherBaseToLidarOffsets.end() ? iso_identity() : it->second;
}

// Slam.cxx:230-344 with several input frames; CheckFrames :709-743
void Slam::AddFrames(const std::vector<const std::vector<Point>*>& frames, const std::vector<uint64_t>& stampsUs)
{
  Tick total;
  Times = StageTimes();
  bool allFramesEmpty = true;
  for (auto* f : frames)
    if (f && !f->empty()) allFramesEmpty = false;
  if (frames.empty() || allFramesEmpty)
    return;
  if (stampsUs[0] == CurrentStamp)
    return;
  CurrentFrames.clear();
  static const std::vector<Point> none;
  for (size_t i = 0; i < frames.size(); ++i) CurrentFrames.push_back({frames[i] ? frames[i] : &none, stampsUs[i]});
  CurrentFrame = CurrentFrames[0].cloud;
  CurrentStamp = stampsUs[0];
  CurrentTime = StampToSec(stampsUs[0]);

  {
    Tick t;
    // Slam::ExtractKeypoints (Slam.cxx:746-810)
    for (int k = 0; k < 3; ++k) PreviousRawKeypoints[k] = std::move(CurrentRawKeypoints[k]);
    std::vector<std::vector<Point>> extracted[3];  // kept alive for the aggregation below
    std::vector<uint64_t> extractedStamps;
    for (const StampedCloud& frame : CurrentFrames)
    {
      if (frame.cloud->empty())
        continue;
      int lidarDevice = frame.cloud->front().device_id;
      Extractor* ke = nullptr;
      if (lidarDevice == 0) ke = &KeyPointsExtractor;
      else if (OtherExtractors.count(lidarDevice)) ke = &OtherExtractors[lidarDevice];
      else if (OtherExtractors.empty()) ke = &KeyPointsExtractor;  // single extractor: the default one stands in
      else continue;                                              // otherwise the frame is ignored
      ke->P.NbThreads = NbThreads;
      ke->ComputeKeyPoints(*frame.cloud);
      for (int k = 0; k < 3; ++k) extracted[k].push_back(ke->Keypoints[k]);
      extractedStamps.push_back(frame.stampUs);
    }
    for (int k = 0; k < 3; ++k)
    {
      if (!UseKeypoints[k]) { CurrentRawKeypoints[k].clear(); continue; }
      std::vector<StampedCloud> clouds;
      for (size_t i = 0; i < extracted[k].size(); ++i) clouds.push_back({&extracted[k][i], extractedStamps[i]});
      CurrentRawKeypoints[k] = AggregateFrames(clouds, false);
    }
    Times.extract = t.Stop();
  }
  ComputeEgoMotion();
  Localization();
  if (OverlapSamplingRatio > 0) EstimateOverlap();
  if (TimeWindowDuration > 0) CheckMotionLimits();
  if (MapUpdate == MappingMode::ADD_KPTS_TO_FIXED_MAP || MapUpdate == MappingMode::UPDATE)
  {
    Tick t;
    UpdateMapsUsingTworld();
    Times.maps = t.Stop();
  }
  LogCurrentFrameState(CurrentTime);
  NbrFrameProcessed++;
  Times.total = total.Stop();
}

// Slam.cxx:1512-1578
std::vector<Point> Slam::AggregateFrames(const std::vector<StampedCloud>& frames, bool worldCoordinates) const
{
  std::vector<Point> aggregated;
  const uint64_t aggregatedStamp = CurrentStamp;  // CurrentFrames[0]->header.stamp
  for (const StampedCloud& frame : frames)
  {
    if (frame.cloud->empty())
      continue;
    const size_t startIdx = aggregated.size();
    aggregated.insert(aggregated.end(), frame.cloud->begin(), frame.cloud->end());
    const size_t endIdx = aggregated.size();
    const double timeOffset = StampToSec(frame.stampUs) - StampToSec(aggregatedStamp);
    const Iso baseToLidar = GetBaseToLidarOffset(frame.cloud->front().device_id);
    if (worldCoordinates && Undistortion)
    {
      Interpolator interp = WithinFrameMotion;
      interp.SetTransforms(iso_mul(iso_mul(Tworld, WithinFrameMotion.GetH0()), baseToLidar),
                           iso_mul(iso_mul(Tworld, WithinFrameMotion.GetH1()), baseToLidar));
      for (size_t i = startIdx; i < endIdx; ++i)
      {
        aggregated[i].time += timeOffset;
        transform_point(aggregated[i], interp(aggregated[i].time));
      }
    }
    else
    {
      const Iso tf = worldCoordinates ? iso_mul(Tworld, baseToLidar) : baseToLidar;
      if (iso_is_approx(tf, iso_identity()))
      {
        for (size_t i = startIdx; i < endIdx; ++i) aggregated[i].time += timeOffset;
      }
      else
      {
        for (size_t i = startIdx; i < endIdx; ++i)
        {
          aggregated[i].time += timeOffset;
          transform_point(aggregated[i], tf);
        }
      }
    }
  }
  return aggregated;
}

// Slam.cxx:1512-1578 with worldCoordinates == false
std::vector<Point> Slam::AggregateKeypoints(const std::vector<Point>& kpts) const
{
  std::vector<Point> out = kpts;
  const double timeOffset = 0.;  // single frame: frame stamp == aggregated stamp
  if (iso_is_approx(BaseToLidarOffset, iso_identity()))
  {
    for (auto& p : out) p.time += timeOffset;
  }
  else
  {
    for (auto& p : out)
    {
      p.time += timeOffset;
      transform_point(p, BaseToLidarOffset);
    }
  }
  return out;
}

// Slam.cxx:746-810
void Slam::ExtractKeypoints()
{
  for (int k = 0; k < 3; ++k) PreviousRawKeypoints[k] = std::move(CurrentRawKeypoints[k]);
  KeyPointsExtractor.P.NbThreads = NbThreads;
  KeyPointsExtractor.ComputeKeyPoints(*CurrentFrame);
  for (int k = 0; k < 3; ++k)
  {
    if (UseKeypoints[k])
      CurrentRawKeypoints[k] = AggregateKeypoints(KeyPointsExtractor.Keypoints[k]);
    else
      CurrentRawKeypoints[k].clear();
  }
}

// Slam.cxx:813-972
void Slam::ComputeEgoMotion()
{
  Trelative = iso_identity();
  if (LogTrajectory.size() >= 2 &&
      (EgoMotion == EgoMotionMode::MOTION_EXTRAPOLATION || EgoMotion == EgoMotionMode::MOTION_EXTRAPOLATION_AND_REGISTRATION))
  {
    const double t = StampToSec(CurrentStamp);
    const double t1 = LogTrajectory[LogTrajectory.size() - 1].time;
    const double t0 = LogTrajectory[LogTrajectory.size() - 2].time;
    if (!(std::abs((t - t1) / (t1 - t0)) > MaxExtrapolationRatio))
    {
      Iso next = linear_interpolation(PreviousTworld, Tworld, t, t0, t1);
      Trelative = iso_mul(iso_inverse(Tworld), next);
    }
  }

  if (EgoMotion == EgoMotionMode::REGISTRATION || EgoMotion == EgoMotionMode::MOTION_EXTRAPOLATION_AND_REGISTRATION)
  {
    for (int k : {EDGE, PLANE}) EgoTrees[k].Reset(&PreviousRawKeypoints[k]);
    TotalMatchedKeypoints = 0;

    MatchParams mp;
    mp.NbThreads = NbThreads;
    mp.SingleEdgePerRing = true;
    mp.MaxNeighborsDistance = EgoMotionMaxNeighborsDistance;
    mp.EdgeNbNeighbors = EgoMotionEdgeNbNeighbors;
    mp.EdgeMinNbNeighbors = EgoMotionEdgeMinNbNeighbors;
    mp.EdgeMaxModelError = EgoMotionEdgeMaxModelError;
    mp.PlaneNbNeighbors = EgoMotionPlaneNbNeighbors;
    mp.PlanarityThreshold = EgoMotionPlanarityThreshold;
    mp.PlaneMaxModelError = EgoMotionPlaneMaxModelError;

    for (unsigned icpIter = 0; icpIter < EgoMotionICPMaxIter; ++icpIter)
    {
      Tick ticp;
      double iterRatio = icpIter / static_cast<double>(EgoMotionICPMaxIter - 1);
      mp.SaturationDistance = (1 - iterRatio) * EgoMotionInitSaturationDistance + iterRatio * EgoMotionFinalSaturationDistance;
      KeypointsMatcher matcher(mp, Trelative);
      for (int k : {EDGE, PLANE})
        EgoMotionMatchingResults[k] = matcher.BuildMatchResiduals(CurrentRawKeypoints[k], EgoTrees[k], (Keypoint)k);
      TotalMatchedKeypoints = 0;
      for (int k : {EDGE, PLANE}) TotalMatchedKeypoints += EgoMotionMatchingResults[k].NbMatches();
      Times.ego_icp += ticp.Stop();
      Times.ego_iters++;
      if (TotalMatchedKeypoints < MinNbMatchedKeypoints)
        break;

      Tick tlm;
      LocalOptimizer optimizer;
      optimizer.SetTwoDMode(TwoDMode);
      optimizer.SetPosePrior(Trelative);
      optimizer.SetLMMaxIter(EgoMotionLMMaxIter);
      optimizer.SetNbThreads(NbThreads);
      for (int k : {EDGE, PLANE}) optimizer.AddResiduals(EgoMotionMatchingResults[k].Residuals);
      LMSummary summary = optimizer.Solve();
      Trelative = optimizer.GetOptimizedPose();
      Times.ego_lm += tlm.Stop();
      Times.lm_evals += summary.num_evaluations;
      if (summary.num_successful_steps == 1)
        break;
    }
  }
}

// Slam.cxx:975-1175
void Slam::Localization()
{
  PreviousTworld = Tworld;
  Tworld = iso_mul(PreviousTworld, Trelative);
  for (int k = 0; k < 3; ++k) CurrentUndistortedKeypoints[k] = CurrentRawKeypoints[k];

  if (Undistortion)
  {
    Tick t;
    InitUndistortion();
    RefineUndistortion();
    Times.undistort += t.Stop();
  }

  {
    Tick t;
    for (int k = 0; k < 3; ++k)
    {
      if (UseKeypoints[k] && !LocalMaps[k]->IsSubMapKdTreeValid())
      {
        if (MapUpdate == MappingMode::NONE)
          LocalMaps[k]->BuildSubMapKdTree();
        else
        {
          if (LocalMaps[k]->IsTimeThreshold())
            LocalMaps[k]->ClearOldPoints(CurrentTime);
          std::vector<Point> currWorld = CurrentUndistortedKeypoints[k];
          float mn[3], mx[3];
          for (int i = 0; i < 3; ++i) { mn[i] = std::numeric_limits<float>::max(); mx[i] = -std::numeric_limits<float>::max(); }
          for (auto& p : currWorld)
          {
            transform_point(p, Tworld);
            const float v[3] = {p.x, p.y, p.z};
            for (int i = 0; i < 3; ++i) { mn[i] = std::min(mn[i], v[i]); mx[i] = std::max(mx[i], v[i]); }
          }
          LocalMaps[k]->BuildSubMapKdTree(mn, mx, currWorld.size() / 2);
        }
      }
    }
    Times.submap += t.Stop();
  }

  TotalMatchedKeypoints = 0;
  MatchParams mp;
  mp.NbThreads = NbThreads;
  mp.SingleEdgePerRing = false;
  mp.MaxNeighborsDistance = LocalizationMaxNeighborsDistance;
  mp.EdgeNbNeighbors = LocalizationEdgeNbNeighbors;
  mp.EdgeMinNbNeighbors = LocalizationEdgeMinNbNeighbors;
  mp.EdgeMaxModelError = LocalizationEdgeMaxModelError;
  mp.PlaneNbNeighbors = LocalizationPlaneNbNeighbors;
  mp.PlanarityThreshold = LocalizationPlanarityThreshold;
  mp.PlaneMaxModelError = LocalizationPlaneMaxModelError;
  mp.BlobNbNeighbors = LocalizationBlobNbNeighbors;

  for (unsigned icpIter = 0; icpIter < LocalizationICPMaxIter; ++icpIter)
  {
    Tick ticp;
    double iterRatio = icpIter / static_cast<double>(LocalizationICPMaxIter - 1);
    mp.SaturationDistance = (1 - iterRatio) * LocalizationInitSaturationDistance + iterRatio * LocalizationFinalSaturationDistance;
    KeypointsMatcher matcher(mp, Tworld);
    for (int k = 0; k < 3; ++k)
      LocalizationMatchingResults[k] = matcher.BuildMatchResiduals(CurrentUndistortedKeypoints[k], LocalMaps[k]->GetSubMapKdTree(), (Keypoint)k);
    TotalMatchedKeypoints = 0;
    for (int k = 0; k < 3; ++k) TotalMatchedKeypoints += LocalizationMatchingResults[k].NbMatches();
    Times.loc_icp += ticp.Stop();
    Times.loc_iters++;

    if (TotalMatchedKeypoints < MinNbMatchedKeypoints)
    {
      Trelative = iso_identity();
      Tworld = PreviousTworld;
      if (Undistortion)
        WithinFrameMotion.SetTransforms(iso_identity(), iso_identity());
      break;
    }

    Tick tlm;
    LocalOptimizer optimizer;
    optimizer.SetTwoDMode(TwoDMode);
    optimizer.SetPosePrior(Tworld);
    optimizer.SetLMMaxIter(LocalizationLMMaxIter);
    optimizer.SetNbThreads(NbThreads);
    for (int k = 0; k < 3; ++k) optimizer.AddResiduals(LocalizationMatchingResults[k].Residuals);
    LMSummary summary = optimizer.Solve();
    Times.lm_evals += summary.num_evaluations;
    Tworld = optimizer.GetOptimizedPose();
    Trelative = iso_mul(iso_inverse(PreviousTworld), Tworld);
    if (Undistortion == UNDIST_REFINED)
      RefineUndistortion();
    Times.loc_lm += tlm.Stop();

    if ((summary.num_successful_steps == 1) || (icpIter == LocalizationICPMaxIter - 1))
    {
      LocalizationUncertainty = optimizer.EstimateRegistrationError();
      break;
    }
  }
}

// Slam.cxx:1178-1222
void Slam::UpdateMapsUsingTworld()
{
  Iso motionSinceLastKf = iso_mul(iso_inverse(KfLastPose), Tworld);
  double transSinceLastKf = std::sqrt((motionSinceLastKf.t[0] * motionSinceLastKf.t[0] + motionSinceLastKf.t[1] * motionSinceLastKf.t[1]) +
                                      motionSinceLastKf.t[2] * motionSinceLastKf.t[2]);
  // Eigen::AngleAxisd(R).angle(): 2 atan2(|q.vec|, |q.w|)
  Quat q = quat_from_matrix(motionSinceLastKf.R);
  double n = std::sqrt((q.x * q.x + q.y * q.y) + q.z * q.z);
  double rotSinceLastKf = (n != 0.) ? 2. * t_atan2(n, std::abs(q.w)) : 0.;

  constexpr double MIN_KF_NB = 10.;
  double thresholdCoef = std::min(KfCounter / MIN_KF_NB, 1.);
  unsigned nbMapKpts = 0;
  for (int k = 0; k < 3; ++k) nbMapKpts += LocalMaps[k]->Size();
  bool isNewKeyFrame = nbMapKpts < MinNbMatchedKeypoints * 10 || transSinceLastKf >= thresholdCoef * KfDistanceThreshold ||
                       rotSinceLastKf >= (thresholdCoef * KfAngleThreshold) / 180. * M_PI;
  if (!isNewKeyFrame)
    return;
  KfCounter++;
  KfLastPose = Tworld;
  for (int k = 0; k < 3; ++k)
  {
    CurrentWorldKeypoints[k] = CurrentUndistortedKeypoints[k];
    for (auto& p : CurrentWorldKeypoints[k]) transform_point(p, Tworld);
  }
  for (int k = 0; k < 3; ++k)
    if (UseKeypoints[k])
      LocalMaps[k]->Add(CurrentWorldKeypoints[k], false, CurrentTime);
}

// Slam.cxx:1225-1264 (poses and covariances; the keypoints log belongs to the pose-graph optimisation)
void Slam::LogCurrentFrameState(double time)
{
  if (LoggingTimeout != 0.)
  {
    LogTrajectory.push_back({Tworld, time});
    std::array<double, 36> c;
    for (int i = 0; i < 36; ++i) c[i] = LocalizationUncertainty.Covariance[i];
    LogCovariances.push_back(c);
    if (LoggingTimeout > 0)
      while (time - LogTrajectory.front().time > LoggingTimeout && LogTrajectory.size() > 2)
      {
        LogTrajectory.pop_front();
        LogCovariances.pop_front();
      }
  }
  else
  {
    LogTrajectory.push_back({Tworld, time});
    while (LogTrajectory.size() > 2) LogTrajectory.pop_front();
  }
}

// Slam.cxx:555-590
Iso Slam::GetLatencyCompensatedWorldTransform() const
{
  const size_t n = LogTrajectory.size();
  if (n == 0) return iso_identity();
  if (n == 1) return LogTrajectory.back().pose;
  const StampedPose& previous = LogTrajectory[n - 2];
  const StampedPose& current = LogTrajectory[n - 1];
  if (std::abs(current.time - previous.time) < 1e-6) return current.pose;
  if (std::abs(Latency / (current.time - previous.time)) > MaxExtrapolationRatio) return current.pose;
  return linear_interpolation(previous.pose, current.pose, current.time + Latency, previous.time, current.time);
}

// Slam.cxx:490-501
void Slam::SetWorldTransformFromGuess(const Iso& guess)
{
  Tworld = guess;
  PreviousTworld = Tworld;
  for (int k = 0; k < 3; ++k) CurrentRawKeypoints[k].clear();
}

// Slam.cxx:610-633; order: ego edges, ego planes, loc edges, planes, blobs, position error, orientation error,
// overlap, comply motion limits, latency
void Slam::GetDebugInformation(double out[10]) const
{
  for (int k = 0; k < 2; ++k) out[k] = EgoMotionMatchingResults[k].NbMatches();
  for (int k = 0; k < 3; ++k) out[2 + k] = LocalizationMatchingResults[k].NbMatches();
  out[5] = LocalizationUncertainty.PositionError;
  out[6] = LocalizationUncertainty.OrientationError;
  out[7] = OverlapEstimation;
  out[8] = ComplyMotionLimits ? 1. : 0.;
  out[9] = Latency;
}

// Slam.cxx:1391-1484
void Slam::CheckMotionLimits()
{
  int nPoses = (int)LogTrajectory.size();
  if (nPoses == 0)
    return;
  double currentTimeStamp = StampToSec(CurrentStamp);
  double deltaTime = currentTimeStamp - LogTrajectory.back().time;
  double nextDeltaTime = FLT_MAX;
  int startIndex = nPoses - 1;
  if (deltaTime < TimeWindowDuration)
  {
    while (startIndex >= 0)
    {
      deltaTime = nextDeltaTime;
      nextDeltaTime = currentTimeStamp - LogTrajectory[startIndex].time;
      if (nextDeltaTime >= TimeWindowDuration)
        break;
      --startIndex;
    }
    if (startIndex < 0)
      startIndex = 0;
    else if (std::abs(deltaTime - TimeWindowDuration) < std::abs(nextDeltaTime - TimeWindowDuration))
      ++startIndex;
    deltaTime = currentTimeStamp - LogTrajectory[startIndex].time;
  }
  ComplyMotionLimits = true;
  Iso TWindow = iso_mul(iso_inverse(LogTrajectory[startIndex].pose), Tworld);
  // Eigen::AngleAxisd(R).angle(): 2 atan2(|q.vec|, |q.w|)
  Quat q = quat_from_matrix(TWindow.R);
  double qn = std::sqrt((q.x * q.x + q.y * q.y) + q.z * q.z);
  float angle = (float)((qn != 0.) ? 2. * t_atan2(qn, std::abs(q.w)) : 0.);
  if (angle > M_PI)
    angle = (float)(2 * M_PI - angle);
  angle = (float)(angle / M_PI * 180.);  // Utils::Rad2Deg (Utilities.h:150-153)
  float distance = (float)std::sqrt((TWindow.t[0] * TWindow.t[0] + TWindow.t[1] * TWindow.t[1]) + TWindow.t[2] * TWindow.t[2]);
  float velocity[2] = {(float)(distance / deltaTime), (float)(angle / deltaTime)};
  if (NbrFrameProcessed >= 2)
  {
    bool comply = true;
    for (int i = 0; i < 2; ++i)
    {
      // Eigen::Array2f / double: the scalar is converted to the array's type first
      float acceleration = (velocity[i] - PreviousVelocity[i]) / (float)deltaTime;
      comply = comply && velocity[i] < VelocityLimits[i] && std::abs(acceleration) < AccelerationLimits[i];
    }
    ComplyMotionLimits = comply;
  }
  PreviousVelocity[0] = velocity[0];
  PreviousVelocity[1] = velocity[1];
}

// Slam.cxx:1271-1285
Iso Slam::InterpolateScanPose(double time)
{
  if (LogTrajectory.empty())
    return Tworld;
  const double prevPoseTime = LogTrajectory.back().time;
  const double currPoseTime = StampToSec(CurrentStamp);
  if (std::abs(time / (currPoseTime - prevPoseTime)) > MaxExtrapolationRatio)
    return Tworld;
  return linear_interpolation(PreviousTworld, Tworld, currPoseTime + time, prevPoseTime, currPoseTime);
}

// Slam.cxx:1288-1319
void Slam::InitUndistortion()
{
  double frameFirstTime = std::numeric_limits<double>::max();
  double frameLastTime = std::numeric_limits<double>::lowest();
  for (int k = 0; k < 3; ++k)
    for (const auto& p : CurrentUndistortedKeypoints[k])
    {
      frameFirstTime = std::min(frameFirstTime, p.time);
      frameLastTime = std::max(frameLastTime, p.time);
    }
  WithinFrameMotion.SetTimes(frameFirstTime, frameLastTime);
  WithinFrameMotion.SetTransforms(iso_identity(), iso_identity());
  if (WithinFrameMotion.GetTimeRange() < 1e-6)
    WithinFrameMotion.SetTimes(0., 0.);
}

// Slam.cxx:1322-1352
void Slam::RefineUndistortion()
{
  Iso previousBaseBegin = WithinFrameMotion.GetH0();
  Iso previousBaseEnd = WithinFrameMotion.GetH1();
  Iso worldToBaseBegin = InterpolateScanPose(WithinFrameMotion.Time0);
  Iso worldToBaseEnd = InterpolateScanPose(WithinFrameMotion.Time1);
  Iso baseToWorld = iso_inverse(Tworld);
  Iso newBaseBegin = iso_mul(baseToWorld, worldToBaseBegin);
  Iso newBaseEnd = iso_mul(baseToWorld, worldToBaseEnd);
  WithinFrameMotion.SetTransforms(newBaseBegin, newBaseEnd);

  Interpolator interp = WithinFrameMotion;
  interp.SetTransforms(iso_mul(newBaseBegin, iso_inverse(previousBaseBegin)), iso_mul(newBaseEnd, iso_inverse(previousBaseEnd)));
  for (int k = 0; k < 3; ++k)
  {
    int nb = CurrentUndistortedKeypoints[k].size();
    #pragma omp parallel for num_threads(NbThreads)
    for (int i = 0; i < nb; ++i)
    {
      Point& p = CurrentUndistortedKeypoints[k][i];
      transform_point(p, interp(p.time));
    }
  }
}

// Slam::EstimateOverlap (Slam.cxx:1370-1388) + Confidence::LCPEstimator (ConfidenceEstimators.cxx:27-65)
void Slam::EstimateOverlap()
{
  const std::vector<Point> cloud = GetRegisteredFrame();
  const float ratio = OverlapSamplingRatio;
  const int nbPoints = (int)(cloud.size() * ratio);
  bool any = false;
  for (int k = 0; k < 3; ++k) any = any || (UseKeypoints[k] && LocalMaps[k]->IsSubMapKdTreeValid());
  if (nbPoints == 0 || !any) { OverlapEstimation = -1.f; return; }
  float lcp = 0.f;
  // the reference reduces with OpenMP (summation order not defined); sequential here
  for (int n = 0; n < nbPoints; ++n)
  {
    const Point& point = cloud[(size_t)(n / ratio)];
    float bestProba = 0.f;
    for (int k = 0; k < 3; ++k)
    {
      if (!(UseKeypoints[k] && LocalMaps[k]->IsSubMapKdTreeValid())) continue;
      int nnIndex[2];
      float nnSqDist[2];
      const float q[3] = {point.x, point.y, point.z};
      if (LocalMaps[k]->GetSubMapKdTree().KnnSearch(q, 1, nnIndex, nnSqDist))
      {
        const float sqLCPThreshold = (float)std::pow(LocalMaps[k]->GetLeafSize() / 3.f, 2);
        const float currentProba = std::exp(-nnSqDist[0] / (2.f * sqLCPThreshold));
        if (currentProba > bestProba) bestProba = currentProba;
      }
    }
    lcp += bestProba;
  }
  OverlapEstimation = lcp / nbPoints;
}

// Slam.cxx:660-667 + 1512-1578 with worldCoordinates == true
std::vector<Point> Slam::GetRegisteredFrame()
{
  std::vector<Point> out;
  if (!CurrentFrame) return out;
  if (!CurrentFrames.empty()) return AggregateFrames(CurrentFrames, true);
  out = *CurrentFrame;
  if (Undistortion)
  {
    Interpolator interp = WithinFrameMotion;
    interp.SetTransforms(iso_mul(iso_mul(Tworld, WithinFrameMotion.GetH0()), BaseToLidarOffset),
                         iso_mul(iso_mul(Tworld, WithinFrameMotion.GetH1()), BaseToLidarOffset));
    int nb = out.size();
    #pragma omp parallel for num_threads(NbThreads)
    for (int i = 0; i < nb; ++i) transform_point(out[i], interp(out[i].time));
  }
  else
  {
    Iso tf = iso_mul(Tworld, BaseToLidarOffset);
    if (!iso_is_approx(tf, iso_identity()))
    {
      int nb = out.size();
      #pragma omp parallel for num_threads(NbThreads)
      for (int i = 0; i < nb; ++i) transform_point(out[i], tf);
    }
  }
  return out;
}

}  // namespace orc
