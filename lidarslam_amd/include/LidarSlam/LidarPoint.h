// LidarSlam/LidarPoint.h -- the reference's point type (slam_lib/include/LidarSlam/LidarPoint.h:31-77) and, where PCL
// is not installed, a minimal pcl::PointCloud with the members the reference touches.  Part of the source-level
// mirror of the reference's public API (see Slam.h).
#pragma once
#include <cstdint>
#include <memory>
#include <string>
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

}  // namespace LidarSlam
