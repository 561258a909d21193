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
// Not mirrored (outside the hot path): pose-graph optimisation, GPS calibration, wheel
// odometry / IMU constraints, PCD map IO, keypoint logging, overlap / motion-limit estimators.
#pragma once
#include <array>
#include <cstdint>
#include <cstring>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <vector>
#include "lidarslam_amd.h"

#if defined(__has_include)
#if __has_include(<pcl/point_cloud.h>)
#include <pcl/point_cloud.h>
#define LSA_HAVE_PCL 1
#endif
#if __has_include(<Eigen/Geometry>)
#include <Eigen/Geometry>
#define LSA_HAVE_EIGEN 1
#endif
#endif

#ifndef LSA_HAVE_PCL
namespace pcl
{
struct PCLHeader
{
  std::uint32_t seq = 0;
  std::uint64_t stamp = 0;  // microseconds
  std::string frame_id;
};
template <typename PointT> class PointCloud
{
public:
  using Ptr = std::shared_ptr<PointCloud<PointT>>;
  using ConstPtr = std::shared_ptr<const PointCloud<PointT>>;
  using PointType = PointT;
  PCLHeader header;
  std::vector<PointT> points;
  bool is_dense = true;
  std::size_t size() const { return points.size(); }
  bool empty() const { return points.empty(); }
  void reserve(std::size_t n) { points.reserve(n); }
  void clear() { points.clear(); }
  void push_back(const PointT& p) { points.push_back(p); }
  PointT& operator[](std::size_t i) { return points[i]; }
  const PointT& operator[](std::size_t i) const { return points[i]; }
  PointT& at(std::size_t i) { return points.at(i); }
  const PointT& at(std::size_t i) const { return points.at(i); }
  PointT& front() { return points.front(); }
  const PointT& front() const { return points.front(); }
  PointT& back() { return points.back(); }
  const PointT& back() const { return points.back(); }
  typename std::vector<PointT>::iterator begin() { return points.begin(); }
  typename std::vector<PointT>::iterator end() { return points.end(); }
  typename std::vector<PointT>::const_iterator begin() const { return points.begin(); }
  typename std::vector<PointT>::const_iterator end() const { return points.end(); }
};
}  // namespace pcl
#endif

namespace LidarSlam
{

// slam_lib/include/LidarSlam/LidarPoint.h:31-64 -- byte-compatible with lsa_point_t
struct LidarPoint
{
  union
  {
    float data[4];
    struct { float x, y, z; };
  };
  double time = 0.;
  float intensity = 0.f;
  std::uint16_t laser_id = 0;
  std::uint8_t device_id = 0;
  std::uint8_t label = 0;
  LidarPoint() : data{0.f, 0.f, 0.f, 1.f} {}
};
static_assert(sizeof(LidarPoint) == sizeof(lsa_point_t), "LidarPoint must stay 32 bytes");

// slam_lib/include/LidarSlam/Enums.h
enum Keypoint { EDGE = 0, PLANE = 1, BLOB = 2, nKeypointTypes };
static const std::vector<Keypoint> KeypointTypes = {EDGE, PLANE, BLOB};
static const std::map<Keypoint, std::string> KeypointTypeNames = {{EDGE, "edge"}, {PLANE, "plane"}, {BLOB, "blob"}};
enum UndistortionMode { NONE = 0, ONCE = 1, REFINED = 2 };
enum class EgoMotionMode { NONE = 0, MOTION_EXTRAPOLATION = 1, REGISTRATION = 2, MOTION_EXTRAPOLATION_AND_REGISTRATION = 3 };
enum class MappingMode { NONE = 0, ADD_KPTS_TO_FIXED_MAP = 1, UPDATE = 2 };
enum class SamplingMode { FIRST = 0, LAST = 1, MAX_INTENSITY = 2, CENTER_POINT = 3, CENTROID = 4 };

// slam_lib/include/LidarSlam/Transform.h:28-79
struct Transform
{
  std::array<double, 16> matrix{{1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1}};  // row-major
  double time = 0.;
  std::string frameid;
  static Transform Identity() { return Transform(); }
  double x() const { return matrix[3]; }
  double y() const { return matrix[7]; }
  double z() const { return matrix[11]; }
  const std::array<double, 16>& GetMatrix() const { return matrix; }
#ifdef LSA_HAVE_EIGEN
  Eigen::Isometry3d GetIsometry() const
  {
    Eigen::Isometry3d iso = Eigen::Isometry3d::Identity();
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 4; ++c) iso(r, c) = matrix[r * 4 + c];
    return iso;
  }
#endif
};

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
  void AddFrames(const std::vector<PointCloud::Ptr>& frames)
  {
    // one LiDAR device per call in this round (the reference aggregates several, Slam.cxx:753-801)
    if (frames.empty() || !frames[0]) return;
    const PointCloud& pc = *frames[0];
    this->CurrentStamp = pc.header.stamp;
    const int rc = lsa_slam_add_frame(this->Handle, reinterpret_cast<const lsa_point_t*>(pc.points.data()), static_cast<int>(pc.size()),
                                      pc.header.stamp, pc.header.seq);
    if (rc < 0) this->LastError = lsa_slam_last_error(this->Handle);  // like the reference: report, keep the previous pose
  }
  Transform GetWorldTransform() const
  {
    Transform t;
    lsa_slam_get_world_transform(this->Handle, t.matrix.data(), &t.time);
    t.frameid = this->WorldFrameId;
    return t;
  }
  std::array<double, 36> GetTransformCovariance() const
  {
    std::array<double, 36> c{};
    lsa_slam_get_covariance(this->Handle, c.data());
    return c;
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
  double GetLatency() const
  {
    double s[16];
    lsa_slam_get_stats(this->Handle, s);
    return s[0];
  }
  // Slam::GetDebugArray (Slam.cxx:635-657); needs SetKeepMatchDebug(true)
  std::unordered_map<std::string, std::vector<double>> GetDebugArray() const
  {
    std::unordered_map<std::string, std::vector<double>> map;
    for (int loc = 0; loc < 2; ++loc)
      for (Keypoint k : KeypointTypes)
      {
        if (!loc && k == BLOB) continue;
        std::vector<std::uint8_t> st(1 << 20);
        std::vector<double> w(1 << 20);
        const int n = lsa_slam_get_match_status(this->Handle, loc, k, st.data(), w.data(), static_cast<int>(st.size()));
        const std::string prefix = std::string(loc ? "Localization: " : "EgoMotion: ") + KeypointTypeNames.at(k);
        map[prefix + " matches"] = std::vector<double>(st.begin(), st.begin() + (n > 0 ? n : 0));
        w.resize(n > 0 ? n : 0);
        map[prefix + " weights"] = w;
      }
    return map;
  }
  const std::string& GetLastError() const { return this->LastError; }

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
  template <typename F> PointCloud::Ptr Fetch(F f, const std::string& frame)
  {
    PointCloud::Ptr pc(new PointCloud);
    std::vector<lsa_point_t> tmp(1 << 19);
    int n = f(tmp.data(), static_cast<int>(tmp.size()));
    if (n == static_cast<int>(tmp.size()))
    {
      tmp.resize(1 << 23);
      n = f(tmp.data(), static_cast<int>(tmp.size()));
    }
    if (n < 0) n = 0;
    pc->points.resize(n);
    if (n > 0) std::memcpy(static_cast<void*>(pc->points.data()), tmp.data(), static_cast<std::size_t>(n) * sizeof(lsa_point_t));
    pc->header.stamp = this->CurrentStamp;
    pc->header.frame_id = frame;
    return pc;
  }

  lsa_slam* Handle = nullptr;
  std::uint64_t CurrentStamp = 0;
  std::string WorldFrameId = "world", BaseFrameId = "base", LastError;
  int Verbosity = 0;
};

#undef LSA_SLAM_PARAM
#undef LSA_SLAM_ENUM_PARAM

}  // namespace LidarSlam
