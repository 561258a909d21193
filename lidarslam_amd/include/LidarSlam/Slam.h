// LidarSlam/Slam.h -- source-level mirror of the reference's public C++ API on top of the C ABI.
//
// Same namespace, class and method names as slam_lib/include/LidarSlam/Slam.h:98-394 for the part of
// the API that drives the per-frame hot path, so that a caller written against the reference
// (ros_wrapping/lidar_slam/src/LidarSlamNode.cxx:173, paraview_wrapping/.../vtkSlam.cxx:217) compiles
// against this header and links liblidarslam_amd.so instead of libLidarSlam.  Header only: every
// method forwards to include/lidarslam_amd.h.
//
// PCL / Eigen: when <pcl/point_cloud.h> is available the real pcl::PointCloud is used; otherwise a
// minimal shim with the members the reference touches (points, header{stamp,frame_id,seq}, size, empty,
// push_back, front, back, reserve, clear, Ptr) is provided so that this header is usable in images
// without PCL (such as the build container).  Poses are exchanged as row-major 4x4 doubles; with Eigen
// present Transform::GetIsometry() is offered as well.
//
// Outside the hot path and therefore present as signature-compatible "not supported on this build"
// members only (they warn once and change nothing): pose-graph optimisation, PCD map IO, wheel odometry / IMU
// constraints, keypoint logging storage.
#pragma once
#include <algorithm>
#include <array>
#include <cmath>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <map>
#include <memory>
#include <set>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <vector>
#include "lidarslam_amd.h"

#include "Enums.h"
#include "LidarPoint.h"
#include "SpinningSensorKeypointExtractor.h"

namespace LidarSlam
{

// slam_lib/include/LidarSlam/Transform.h:25-79.  The pose is kept as a row-major 4x4 (what the C ABI speaks); with
// Eigen present (LSA_HAVE_EIGEN) the reference's Eigen constructors and accessors are offered with their own types.
struct Transform
{
  std::array<double, 16> matrix{{1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1}};  // row-major
  double time = 0.;
  std::string frameid;

  Transform() = default;
  // Euler angles ZYX convention (R = Rz(rz) Ry(ry) Rx(rx)), as Utils::XYZRPYtoIsometry (slam_lib/src/Utilities.cxx:62-69)
  Transform(double x, double y, double z, double rx, double ry, double rz, double t = 0., const std::string& frame = "") : time(t), frameid(frame)
  {
    const double cx = std::cos(rx), sx = std::sin(rx), cy = std::cos(ry), sy = std::sin(ry), cz = std::cos(rz), sz = std::sin(rz);
    matrix = {{cy * cz, sx * sy * cz - cx * sz, cx * sy * cz + sx * sz, x,
               cy * sz, sx * sy * sz + cx * cz, cx * sy * sz - sx * cz, y,
               -sy, sx * cy, cx * cy, z,
               0, 0, 0, 1}};
  }
  explicit Transform(const std::array<double, 16>& rowMajor, double t = 0., const std::string& frame = "") : matrix(rowMajor), time(t), frameid(frame) {}
  static Transform Identity() { return Transform(); }

  double& x() { return matrix[3]; }
  double& y() { return matrix[7]; }
  double& z() { return matrix[11]; }
  double x() const { return matrix[3]; }
  double y() const { return matrix[7]; }
  double z() const { return matrix[11]; }
  const std::array<double, 16>& GetMatrixArray() const { return matrix; }
#ifdef LSA_HAVE_EIGEN
  Transform(const Eigen::Matrix<double, 6, 1>& xyzrpy, double t = 0., const std::string& frame = "")
    : Transform(xyzrpy(0), xyzrpy(1), xyzrpy(2), xyzrpy(3), xyzrpy(4), xyzrpy(5), t, frame) {}
  Transform(const Eigen::Vector3d& trans, const Eigen::Vector3d& rpy, double t = 0., const std::string& frame = "")
    : Transform(trans(0), trans(1), trans(2), rpy(0), rpy(1), rpy(2), t, frame) {}
  Transform(const Eigen::Isometry3d& transform, double t = 0., const std::string& frame = "") : time(t), frameid(frame) { this->SetIsometry(transform); }
  Transform(const Eigen::Translation3d& trans, const Eigen::Quaterniond& rot, double t = 0., const std::string& frame = "") : time(t), frameid(frame)
  {
    this->SetIsometry(Eigen::Isometry3d(trans * rot.normalized()));
  }
  void SetIsometry(const Eigen::Isometry3d& isometry)
  {
    for (int r = 0; r < 4; ++r)
      for (int c = 0; c < 4; ++c) matrix[r * 4 + c] = isometry.matrix()(r, c);
  }
  Eigen::Isometry3d GetIsometry() const
  {
    Eigen::Isometry3d iso = Eigen::Isometry3d::Identity();
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 4; ++c) iso(r, c) = matrix[r * 4 + c];
    return iso;
  }
  Eigen::Vector3d GetPosition() const { return Eigen::Vector3d(matrix[3], matrix[7], matrix[11]); }
  Eigen::Translation3d GetTranslation() const { return Eigen::Translation3d(this->GetPosition()); }
  Eigen::Quaterniond GetRotation() const { return Eigen::Quaterniond(this->GetIsometry().linear()); }
  Eigen::Matrix4d GetMatrix() const { return this->GetIsometry().matrix(); }
#else
  // without Eigen: position as an array, the matrix as the row-major array
  std::array<double, 3> GetPosition() const { return {{matrix[3], matrix[7], matrix[11]}}; }
  const std::array<double, 16>& GetMatrix() const { return matrix; }
#endif
};

// slam_lib/include/LidarSlam/PointCloudStorage.h:60-75 (the values callers pass to SaveMapsToPCD / SetLoggingStorage)
enum PCDFormat { ASCII = 0, BINARY = 1, BINARY_COMPRESSED = 2 };
enum PointCloudStorageType { PCL_CLOUD = 0, OCTREE_COMPRESSED = 1, PCD_ASCII = 2, PCD_BINARY = 3, PCD_BINARY_COMPRESSED = 4 };

// slam_lib/include/LidarSlam/SensorConstraints.h:13-23 (external sensors: accepted, not used -- see the Slam methods)
namespace SensorConstraints
{
struct WheelOdomMeasurement
{
  double Time = 0.;
  double Distance = 0.;
};
struct GravityMeasurement
{
  double Time = 0.;
#ifdef LSA_HAVE_EIGEN
  Eigen::Vector3d Acceleration = Eigen::Vector3d::Zero();
#else
  std::array<double, 3> Acceleration{{0., 0., 0.}};
#endif
};
}  // namespace SensorConstraints

#define LSA_SLAM_PARAM(name, type)                                                          \
  void Set##name(type v) { this->SetParam(#name, static_cast<double>(v)); }                 \
  type Get##name() const { return static_cast<type>(this->GetParam(#name)); }
#define LSA_SLAM_ENUM_PARAM(name, type)                                                     \
  void Set##name(type v) { this->SetParam(#name, static_cast<double>(static_cast<int>(v))); } \
  type Get##name() const { return static_cast<type>(static_cast<int>(this->GetParam(#name))); }

class Slam
{
public:
  using Point = LidarPoint;
  using PointCloud = pcl::PointCloud<Point>;

  // Throws when no HIP device is usable: the device path is the only path.
  explicit Slam(int device = 0)
  {
    if (lsa_slam_create(device, &this->Handle) != LSA_OK)
      throw std::runtime_error("LidarSlam::Slam: no usable HIP device (liblidarslam_amd has no CPU fallback)");
  }
  ~Slam() { lsa_slam_destroy(this->Handle); }
  Slam(const Slam&) = delete;
  Slam& operator=(const Slam&) = delete;

  void Reset(bool resetLog = true) { lsa_slam_reset(this->Handle, resetLog ? 1 : 0); }

  // ---- main use (Slam.h:111-146)
  void AddFrame(const PointCloud::Ptr& pc) { this->AddFrames({pc}); }
  // one frame per LiDAR device (at most 16): extracted with the device's extractor, merged in BASE (Slam.cxx:753-801)
  void AddFrames(const std::vector<PointCloud::Ptr>& frames)
  {
    if (frames.empty() || !frames[0]) return;
    if (frames.size() > 16)
    {
      this->LastError = "AddFrames: at most 16 frames (LiDAR devices) per call";
      std::fprintf(stderr, "\033[1;31m[ERROR] %s\033[0m\n", this->LastError.c_str());
      return;
    }
    const lsa_point_t* pts[16];
    int n[16];
    std::uint64_t stamps[16];
    std::uint32_t seqs[16];
    for (std::size_t i = 0; i < frames.size(); ++i)
    {
      const bool some = frames[i] && !frames[i]->empty();
      pts[i] = some ? reinterpret_cast<const lsa_point_t*>(frames[i]->points.data()) : nullptr;
      n[i] = some ? static_cast<int>(frames[i]->size()) : 0;
      stamps[i] = frames[i] ? frames[i]->header.stamp : frames[0]->header.stamp;
      seqs[i] = frames[i] ? frames[i]->header.seq : 0;
    }
    this->CurrentStamp = frames[0]->header.stamp;
    const int rc = lsa_slam_add_frames(this->Handle, pts, n, stamps, seqs, static_cast<int>(frames.size()));
    if (rc < 0) this->LastError = lsa_slam_last_error(this->Handle);  // like the reference: report, keep the previous pose
  }
  Transform GetWorldTransform() const
  {
    Transform t;
    lsa_slam_get_world_transform(this->Handle, t.matrix.data(), &t.time);
    t.frameid = this->WorldFrameId;
    return t;
  }
  // the last pose extrapolated by the duration of the last AddFrame (Slam.cxx:555-590)
  Transform GetLatencyCompensatedWorldTransform() const
  {
    Transform t;
    lsa_slam_get_latency_compensated_world_transform(this->Handle, t.matrix.data(), &t.time);
    t.frameid = this->WorldFrameId;
    return t;
  }
  std::array<double, 36> GetTransformCovariance() const
  {
    std::array<double, 36> c{};
    lsa_slam_get_covariance(this->Handle, c.data());
    return c;
  }
  // Slam.h:150-151; the log holds two poses unless SetLoggingTimeout was given something else than 0
  std::vector<Transform> GetTrajectory() const
  {
    const int n = lsa_slam_get_trajectory(this->Handle, nullptr, nullptr, 0);
    std::vector<double> rows(static_cast<std::size_t>(n > 0 ? n : 0) * 17);
    if (n > 0) lsa_slam_get_trajectory(this->Handle, rows.data(), nullptr, n);
    std::vector<Transform> poses(n > 0 ? n : 0);
    for (int i = 0; i < n; ++i)
    {
      std::memcpy(poses[i].matrix.data(), rows.data() + 17 * i, 16 * sizeof(double));
      poses[i].time = rows[17 * i + 16];
      poses[i].frameid = this->WorldFrameId;
    }
    return poses;
  }
  std::vector<std::array<double, 36>> GetCovariances() const
  {
    const int n = lsa_slam_get_trajectory(this->Handle, nullptr, nullptr, 0);
    std::vector<std::array<double, 36>> covs(n > 0 ? n : 0);
    if (n > 0) lsa_slam_get_trajectory(this->Handle, nullptr, covs[0].data(), n);
    return covs;
  }
  // Slam.h:155-158
  PointCloud::Ptr GetMap(Keypoint k, bool clean = false)
  {
    return this->Fetch([&](lsa_point_t* out, int cap) { return std::min(cap, lsa_slam_get_map(this->Handle, k, clean ? 1 : 0, out, cap)); }, this->WorldFrameId);
  }
  PointCloud::Ptr GetTargetSubMap(Keypoint k)
  {
    return this->Fetch([&](lsa_point_t* out, int cap) { return std::min(cap, lsa_slam_get_target_submap(this->Handle, k, out, cap)); }, this->WorldFrameId);
  }
  // Slam.h:189
  void SetWorldTransformFromGuess(const Transform& poseGuess) { lsa_slam_set_world_transform_from_guess(this->Handle, poseGuess.matrix.data()); }
  // Slam.h:176 (same keys)
  std::unordered_map<std::string, double> GetDebugInformation() const
  {
    double v[10] = {};
    lsa_slam_get_debug_information(this->Handle, v);
    return {{"EgoMotion: edges used", v[0]}, {"EgoMotion: planes used", v[1]}, {"Localization: edges used", v[2]},
            {"Localization: planes used", v[3]}, {"Localization: blobs used", v[4]}, {"Localization: position error", v[5]},
            {"Localization: orientation error", v[6]}, {"Confidence: overlap", v[7]}, {"Confidence: comply motion limits", v[8]}};
  }
  // worldCoordinates = false: BASE (undistorted), true: WORLD (Slam.cxx:690-702)
  PointCloud::Ptr GetKeypoints(Keypoint k, bool worldCoordinates = false)
  {
    return this->Fetch([&](lsa_point_t* out, int cap) { return lsa_slam_get_keypoints(this->Handle, k, worldCoordinates ? 1 : 0, out, cap); },
                       worldCoordinates ? this->WorldFrameId : this->BaseFrameId);
  }
  PointCloud::Ptr GetRegisteredFrame()
  {
    return this->Fetch([&](lsa_point_t* out, int cap) { return lsa_slam_get_registered_frame(this->Handle, out, cap); }, this->WorldFrameId);
  }
  unsigned int GetNbrFrameProcessed() const { return static_cast<unsigned int>(this->GetParam("NbrFrameProcessed")); }
  int GetTotalMatchedKeypoints() const { return static_cast<int>(this->GetParam("TotalMatchedKeypoints")); }
  // Confidence estimator (Slam.h: GetOverlapSamplingRatio / SetOverlapSamplingRatio / GetOverlapEstimation)
  float GetOverlapSamplingRatio() const { return static_cast<float>(this->GetParam("OverlapSamplingRatio")); }
  void SetOverlapSamplingRatio(float ratio) { this->SetParam("OverlapSamplingRatio", ratio); }
  float GetOverlapEstimation() const { return static_cast<float>(this->GetParam("OverlapEstimation")); }
  double GetLatency() const { return this->GetParam("Latency"); }
  // Slam::GetDebugArray (Slam.cxx:635-657); needs SetKeepMatchDebug(true)
  std::unordered_map<std::string, std::vector<double>> GetDebugArray() const
  {
    std::unordered_map<std::string, std::vector<double>> map;
    for (int loc = 0; loc < 2; ++loc)
      for (Keypoint k : KeypointTypes)
      {
        if (!loc && k == BLOB) continue;
        // size first (null buffers), then exactly that much
        const int have = lsa_slam_get_match_status(this->Handle, loc, k, nullptr, nullptr, 1 << 30);
        std::vector<std::uint8_t> st(have > 0 ? have : 0);
        std::vector<double> w(st.size());
        const int n = st.empty() ? 0 : lsa_slam_get_match_status(this->Handle, loc, k, st.data(), w.data(), static_cast<int>(st.size()));
        const std::string prefix = std::string(loc ? "Localization: " : "EgoMotion: ") + KeypointTypeNames.at(k);
        map[prefix + " matches"] = std::vector<double>(st.begin(), st.begin() + (n > 0 ? n : 0));
        w.resize(n > 0 ? n : 0);
        map[prefix + " weights"] = w;
      }
    return map;
  }
  const std::string& GetLastError() const { return this->LastError; }

  // ---- outside the scan-matching hot path (pose-graph optimisation, PCD map IO, external sensors, keypoint
  // logging).  Present with the reference's signatures so that LidarSlamNode.cxx / vtkSlam.cxx compile unchanged; each
  // warns once and does nothing, the way the reference itself answers RunPoseGraphOptimization without g2o
  // (slam_lib/src/Slam.cxx:355-366: "SLAM PoseGraphOptimization requires G2O, but it was not found.").
#ifdef LSA_HAVE_EIGEN
  void RunPoseGraphOptimization(const std::vector<Transform>&, const std::vector<std::array<double, 9>>&, Eigen::Isometry3d&, const std::string& = "")
#else
  void RunPoseGraphOptimization(const std::vector<Transform>&, const std::vector<std::array<double, 9>>&, std::array<double, 16>&, const std::string& = "")
#endif
  {
    this->NotSupported("RunPoseGraphOptimization", "pose-graph optimisation (g2o) is not part of this build: maps and trajectory are left unchanged");
  }
  void SaveMapsToPCD(const std::string&, PCDFormat = PCDFormat::BINARY_COMPRESSED, bool = true) const
  {
    this->NotSupported("SaveMapsToPCD", "PCD map IO is not part of this build: use GetMap() and the caller's own writer");
  }
  void LoadMapsFromPCD(const std::string&, bool = true)
  {
    this->NotSupported("LoadMapsFromPCD", "PCD map IO is not part of this build: the maps are left unchanged");
  }
  // keypoint logging feeds the pose-graph optimisation only: the value is remembered, nothing is logged
  void SetLoggingStorage(PointCloudStorageType s) { this->LoggingStorage = s; }
  PointCloudStorageType GetLoggingStorage() const { return this->LoggingStorage; }
  // wheel odometer / IMU constraints: weights and time offset are remembered, measurements are dropped
  void SetWheelOdomWeight(double w) { this->WheelOdomWeight = w; }
  double GetWheelOdomWeight() const { return this->WheelOdomWeight; }
  void SetGravityWeight(double w) { this->GravityWeight = w; }
  double GetGravityWeight() const { return this->GravityWeight; }
  void SetSensorTimeOffset(double t) { this->SensorTimeOffset = t; }
  double GetSensorTimeOffset() const { return this->SensorTimeOffset; }
  void AddGravityMeasurement(const SensorConstraints::GravityMeasurement&)
  {
    this->NotSupported("AddGravityMeasurement", "IMU gravity constraints are not part of this build: the measurement is ignored");
  }
  void AddWheelOdomMeasurement(const SensorConstraints::WheelOdomMeasurement&)
  {
    this->NotSupported("AddWheelOdomMeasurement", "wheel odometry constraints are not part of this build: the measurement is ignored");
  }
  void ClearSensorMeasurements() {}

  // ---- general parameters (Slam.h:201-232)
  void SetNbThreads(int) {}  // OpenMP thread count of the reference: meaningless on the device path
  int GetNbThreads() const { return 1; }
  void SetVerbosity(int v) { this->Verbosity = v; }
  int GetVerbosity() const { return this->Verbosity; }
  LSA_SLAM_PARAM(UseBlobs, bool)
  LSA_SLAM_ENUM_PARAM(EgoMotion, EgoMotionMode)
  LSA_SLAM_ENUM_PARAM(Undistortion, UndistortionMode)
  LSA_SLAM_ENUM_PARAM(MapUpdate, MappingMode)
  LSA_SLAM_PARAM(KeepMatchDebug, bool)
  LSA_SLAM_PARAM(LoggingTimeout, double)
  // Slam.h:239-245: the extraction runs inside the library, so what is taken from the extractor object is its
  // parameters (and its azimuthal resolution when it has one); the object is kept for GetKeyPointsExtractor
  using KeypointExtractorPtr = std::shared_ptr<SpinningSensorKeypointExtractor>;
  void SetKeyPointsExtractor(KeypointExtractorPtr extractor, std::uint8_t deviceId = 0)
  {
    if (!extractor) return;
    const std::pair<const char*, double> values[] = {
      {"NeighborWidth", extractor->GetNeighborWidth()}, {"MinDistanceToSensor", extractor->GetMinDistanceToSensor()},
      {"MinBeamSurfaceAngle", extractor->GetMinBeamSurfaceAngle()}, {"PlaneSinAngleThreshold", extractor->GetPlaneSinAngleThreshold()},
      {"EdgeSinAngleThreshold", extractor->GetEdgeSinAngleThreshold()}, {"EdgeDepthGapThreshold", extractor->GetEdgeDepthGapThreshold()},
      {"EdgeSaliencyThreshold", extractor->GetEdgeSaliencyThreshold()}, {"EdgeIntensityGapThreshold", extractor->GetEdgeIntensityGapThreshold()}};
    for (const auto& v : values) lsa_slam_set_extractor_param(this->Handle, deviceId, v.first, v.second);
    if (extractor->GetAzimuthalResolution() > 0.f)
      lsa_slam_set_extractor_param(this->Handle, deviceId, "AzimuthalResolution", extractor->GetAzimuthalResolution());
    this->KeyPointsExtractors[deviceId] = extractor;
  }
  KeypointExtractorPtr GetKeyPointsExtractor(std::uint8_t deviceId = 0) const
  {
    const auto it = this->KeyPointsExtractors.find(deviceId);
    return it != this->KeyPointsExtractors.end() ? it->second : KeypointExtractorPtr();
  }
  std::map<std::uint8_t, KeypointExtractorPtr> GetKeyPointsExtractors() const { return this->KeyPointsExtractors; }
  void SetKeyPointsExtractors(const std::map<std::uint8_t, KeypointExtractorPtr>& extractors)
  {
    for (const auto& kv : extractors) this->SetKeyPointsExtractor(kv.second, kv.first);
  }
  // Slam.h:249-250 (row-major 4x4)
  void SetBaseToLidarOffset(const std::array<double, 16>& transform, std::uint8_t deviceId = 0)
  {
    lsa_slam_set_base_to_lidar_offset(this->Handle, transform.data(), deviceId);
  }
  std::array<double, 16> GetBaseToLidarOffset(std::uint8_t deviceId = 0) const
  {
    std::array<double, 16> t{{1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1}};
    lsa_slam_get_base_to_lidar_offset(this->Handle, t.data(), deviceId);
    return t;
  }
#ifdef LSA_HAVE_EIGEN
  void SetBaseToLidarOffset(const Eigen::Isometry3d& transform, std::uint8_t deviceId = 0)
  {
    std::array<double, 16> t{};
    for (int r = 0; r < 4; ++r)
      for (int c = 0; c < 4; ++c) t[r * 4 + c] = transform.matrix()(r, c);
    this->SetBaseToLidarOffset(t, deviceId);
  }
#endif
  // confidence estimator on the motion (Slam.h:385-394): {linear, angular} limits
  void SetVelocityLimits(const std::array<float, 2>& l) { this->SetParam("VelocityLimitLinear", l[0]); this->SetParam("VelocityLimitAngular", l[1]); }
  std::array<float, 2> GetVelocityLimits() const { return {{static_cast<float>(this->GetParam("VelocityLimitLinear")), static_cast<float>(this->GetParam("VelocityLimitAngular"))}}; }
  void SetAccelerationLimits(const std::array<float, 2>& l) { this->SetParam("AccelerationLimitLinear", l[0]); this->SetParam("AccelerationLimitAngular", l[1]); }
  std::array<float, 2> GetAccelerationLimits() const { return {{static_cast<float>(this->GetParam("AccelerationLimitLinear")), static_cast<float>(this->GetParam("AccelerationLimitAngular"))}}; }
#ifdef LSA_HAVE_EIGEN
  // the reference's own argument type (Slam.h:385-392)
  void SetVelocityLimits(const Eigen::Array2f& l) { this->SetVelocityLimits(std::array<float, 2>{{l(0), l(1)}}); }
  void SetAccelerationLimits(const Eigen::Array2f& l) { this->SetAccelerationLimits(std::array<float, 2>{{l(0), l(1)}}); }
#endif
  LSA_SLAM_PARAM(TimeWindowDuration, float)
  bool GetComplyMotionLimits() const { return this->GetParam("ComplyMotionLimits") != 0.; }
  void SetBaseFrameId(const std::string& s) { this->BaseFrameId = s; }
  std::string GetBaseFrameId() const { return this->BaseFrameId; }
  void SetWorldFrameId(const std::string& s) { this->WorldFrameId = s; }
  std::string GetWorldFrameId() const { return this->WorldFrameId; }

  // ---- optimisation parameters (Slam.h:262-340)
  LSA_SLAM_PARAM(TwoDMode, bool)
  LSA_SLAM_PARAM(EgoMotionLMMaxIter, unsigned int)
  LSA_SLAM_PARAM(EgoMotionICPMaxIter, unsigned int)
  LSA_SLAM_PARAM(EgoMotionMaxNeighborsDistance, double)
  LSA_SLAM_PARAM(EgoMotionEdgeNbNeighbors, unsigned int)
  LSA_SLAM_PARAM(EgoMotionEdgeMinNbNeighbors, unsigned int)
  LSA_SLAM_PARAM(EgoMotionPlaneNbNeighbors, unsigned int)
  LSA_SLAM_PARAM(EgoMotionPlanarityThreshold, double)
  LSA_SLAM_PARAM(EgoMotionEdgeMaxModelError, double)
  LSA_SLAM_PARAM(EgoMotionPlaneMaxModelError, double)
  LSA_SLAM_PARAM(EgoMotionInitSaturationDistance, double)
  LSA_SLAM_PARAM(EgoMotionFinalSaturationDistance, double)
  LSA_SLAM_PARAM(LocalizationLMMaxIter, unsigned int)
  LSA_SLAM_PARAM(LocalizationICPMaxIter, unsigned int)
  LSA_SLAM_PARAM(LocalizationMaxNeighborsDistance, double)
  LSA_SLAM_PARAM(LocalizationEdgeNbNeighbors, unsigned int)
  LSA_SLAM_PARAM(LocalizationEdgeMinNbNeighbors, unsigned int)
  LSA_SLAM_PARAM(LocalizationPlaneNbNeighbors, unsigned int)
  LSA_SLAM_PARAM(LocalizationPlanarityThreshold, double)
  LSA_SLAM_PARAM(LocalizationEdgeMaxModelError, double)
  LSA_SLAM_PARAM(LocalizationPlaneMaxModelError, double)
  LSA_SLAM_PARAM(LocalizationBlobNbNeighbors, unsigned int)
  LSA_SLAM_PARAM(LocalizationInitSaturationDistance, double)
  LSA_SLAM_PARAM(LocalizationFinalSaturationDistance, double)

  // ---- keyframes and maps (Slam.h:360-378)
  LSA_SLAM_PARAM(KfDistanceThreshold, double)
  LSA_SLAM_PARAM(KfAngleThreshold, double)
  void SetVoxelGridLeafSize(Keypoint k, double size)
  {
    this->SetParam(k == EDGE ? "VoxelGridLeafSizeEdges" : k == PLANE ? "VoxelGridLeafSizePlanes" : "VoxelGridLeafSizeBlobs", size);
  }
  void SetVoxelGridSize(int size) { this->SetParam("VoxelGridSize", size); }
  void SetVoxelGridResolution(double r) { this->SetParam("VoxelGridResolution", r); }
  void SetVoxelGridMinFramesPerVoxel(unsigned int n) { this->SetParam("VoxelGridMinFramesPerVoxel", n); }
  void SetVoxelGridDecayingThreshold(double d) { this->SetParam("VoxelGridDecayingThreshold", d); }
  void SetVoxelGridSamplingMode(Keypoint, SamplingMode sm) { this->SetParam("VoxelGridSamplingMode", static_cast<int>(sm)); }
  SamplingMode GetVoxelGridSamplingMode(Keypoint) { return static_cast<SamplingMode>(static_cast<int>(this->GetParam("VoxelGridSamplingMode"))); }
  double GetVoxelGridDecayingThreshold() { return this->GetParam("VoxelGridDecayingThreshold"); }
  // Slam::ClearMaps (Slam.cxx): empties the three keypoint maps, poses and parameters stay
  void ClearMaps() { lsa_slam_clear_maps(this->Handle); }

  // ---- keypoint extractor parameters (SpinningSensorKeypointExtractor.h:44-70); one extractor (device 0)
  LSA_SLAM_PARAM(NeighborWidth, int)
  LSA_SLAM_PARAM(MinDistanceToSensor, float)
  LSA_SLAM_PARAM(MinBeamSurfaceAngle, float)
  LSA_SLAM_PARAM(PlaneSinAngleThreshold, float)
  LSA_SLAM_PARAM(EdgeSinAngleThreshold, float)
  LSA_SLAM_PARAM(EdgeDepthGapThreshold, float)
  LSA_SLAM_PARAM(EdgeSaliencyThreshold, float)
  LSA_SLAM_PARAM(EdgeIntensityGapThreshold, float)
  LSA_SLAM_PARAM(AzimuthalResolution, float)

  // generic access by the reference's setter name (without "Set")
  void SetParam(const std::string& name, double v)
  {
    if (lsa_slam_set_param(this->Handle, name.c_str(), v) != LSA_OK) throw std::invalid_argument("LidarSlam::Slam: unknown parameter " + name);
  }
  double GetParam(const std::string& name) const
  {
    double v = 0.;
    if (lsa_slam_get_param(this->Handle, name.c_str(), &v) != LSA_OK) throw std::invalid_argument("LidarSlam::Slam: unknown parameter " + name);
    return v;
  }
  lsa_slam* GetHandle() { return this->Handle; }

private:
  // one staging buffer per Slam, kept between calls (the wrappers call the getters every frame: no 16 MB allocation per call)
  template <typename F> PointCloud::Ptr Fetch(F f, const std::string& frame)
  {
    PointCloud::Ptr pc(new PointCloud);
    if (this->Staging.size() < (1u << 19)) this->Staging.resize(1u << 19);
    int n = f(this->Staging.data(), static_cast<int>(this->Staging.size()));
    while (n >= static_cast<int>(this->Staging.size()) && this->Staging.size() < (1u << 26))
    {
      this->Staging.resize(this->Staging.size() * 4);  // it may have been cut short: once more with room
      n = f(this->Staging.data(), static_cast<int>(this->Staging.size()));
    }
    if (n < 0) n = 0;
    n = std::min<int>(n, static_cast<int>(this->Staging.size()));
    pc->points.resize(n);
    if (n > 0) std::memcpy(static_cast<void*>(pc->points.data()), this->Staging.data(), static_cast<std::size_t>(n) * sizeof(lsa_point_t));
    pc->header.stamp = this->CurrentStamp;
    pc->header.frame_id = frame;
    return pc;
  }

  void NotSupported(const char* what, const char* why) const
  {
    if (this->Warned.insert(what).second) std::fprintf(stderr, "\033[1;33m[WARNING] LidarSlam::Slam::%s: %s\033[0m\n", what, why);
  }

  lsa_slam* Handle = nullptr;
  mutable std::set<std::string> Warned;
  std::vector<lsa_point_t> Staging;  // Fetch
  PointCloudStorageType LoggingStorage = PointCloudStorageType::PCL_CLOUD;
  double WheelOdomWeight = 0., GravityWeight = 0., SensorTimeOffset = 0.;
  std::map<std::uint8_t, KeypointExtractorPtr> KeyPointsExtractors;
  std::uint64_t CurrentStamp = 0;
  std::string WorldFrameId = "world", BaseFrameId = "base", LastError;
  int Verbosity = 0;
};

#undef LSA_SLAM_PARAM
#undef LSA_SLAM_ENUM_PARAM

}  // namespace LidarSlam
