// LidarSlam/SpinningSensorKeypointExtractor.h -- source-level mirror of the reference's keypoint extractor
// (slam_lib/include/LidarSlam/SpinningSensorKeypointExtractor.h:38-88) on top of the C ABI: same setters,
// ComputeKeyPoints, GetKeypoints and GetDebugArray, computed by the HIP extraction kernels of its own device
// context.  Header only; there is no CPU fallback (the constructor throws without a usable HIP device).
#pragma once
#include <map>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <vector>
#include "Enums.h"
#include "LidarPoint.h"
#include "lidarslam_amd.h"

namespace LidarSlam
{

#define LSA_SSKE_PARAM(name, type, member)                     \
  type Get##name() const { return this->Params.member; }       \
  void Set##name(type v) { this->Params.member = v; }

class SpinningSensorKeypointExtractor
{
public:
  using Point = LidarPoint;
  using PointCloud = pcl::PointCloud<Point>;

  explicit SpinningSensorKeypointExtractor(int device = 0)
  {
    if (lsa_ctx_create(device, &this->Ctx) != LSA_OK)
      throw std::runtime_error("LidarSlam::SpinningSensorKeypointExtractor: no usable HIP device (no CPU fallback)");
    this->Params = lsa_extract_params_t{4, 1.5f, 10.f, 0.5f, 0.86f, 0.20f, 0.15f, 1.5f, 50.f};  // SSKE.h:125-148
    for (Keypoint k : KeypointTypes) this->Keypoints[k].reset(new PointCloud);
  }
  ~SpinningSensorKeypointExtractor() { lsa_ctx_destroy(this->Ctx); }
  SpinningSensorKeypointExtractor(const SpinningSensorKeypointExtractor&) = delete;
  SpinningSensorKeypointExtractor& operator=(const SpinningSensorKeypointExtractor&) = delete;

  int GetNbThreads() const { return this->NbThreads; }  // OpenMP threads of the reference: no meaning on the device
  void SetNbThreads(int n) { this->NbThreads = n; }
  LSA_SSKE_PARAM(NeighborWidth, int, neighbor_width)
  LSA_SSKE_PARAM(MinDistanceToSensor, float, min_distance_to_sensor)
  LSA_SSKE_PARAM(MinBeamSurfaceAngle, float, min_beam_surface_angle)
  LSA_SSKE_PARAM(PlaneSinAngleThreshold, float, plane_sin_angle_threshold)
  LSA_SSKE_PARAM(EdgeSinAngleThreshold, float, edge_sin_angle_threshold)
  LSA_SSKE_PARAM(EdgeDepthGapThreshold, float, edge_depth_gap_threshold)
  LSA_SSKE_PARAM(EdgeSaliencyThreshold, float, edge_saliency_threshold)
  LSA_SSKE_PARAM(EdgeIntensityGapThreshold, float, edge_intensity_gap_threshold)
  float GetAzimuthalResolution() const { return lsa_get_azimuthal_resolution(this->Ctx); }
  void SetAzimuthalResolution(float r) { lsa_set_azimuthal_resolution(this->Ctx, r); }
  int GetNbLaserRings() const { return lsa_nb_laser_rings(this->Ctx); }

  std::map<Keypoint, PointCloud::Ptr> GetKeypoints() const { return this->Keypoints; }
  PointCloud::Ptr GetKeypoints(Keypoint k) const { return this->Keypoints.at(k); }

  // SSKE.cxx:118-136.  Fresh clouds every frame: callers may hold the previous ones (SSKE.cxx:176-182).
  // On a device error the keypoint clouds stay empty and GetLastError() says why.
  void ComputeKeyPoints(const PointCloud::Ptr& pc)
  {
    this->LastError.clear();
    this->ScanSize = 0;
    for (Keypoint k : KeypointTypes)
    {
      this->Keypoints[k].reset(new PointCloud);
      if (pc) this->Keypoints[k]->header = pc->header;
    }
    if (!pc || pc->empty()) return;
    int counts[3] = {0, 0, 0};
    if (lsa_upload_frame(this->Ctx, reinterpret_cast<const lsa_point_t*>(pc->points.data()), static_cast<int>(pc->size())) != LSA_OK ||
        lsa_extract_keypoints(this->Ctx, &this->Params, counts) != LSA_OK)
    {
      this->LastError = lsa_last_error(this->Ctx);
      return;
    }
    this->ScanSize = static_cast<int>(pc->size());
    for (Keypoint k : KeypointTypes)
    {
      PointCloud& out = *this->Keypoints[k];
      out.points.resize(counts[k]);
      if (counts[k] > 0 &&
          lsa_download_keypoints(this->Ctx, LSA_SET_RAW_CURRENT, k, reinterpret_cast<lsa_point_t*>(out.points.data()), counts[k]) < 0)
      {
        this->LastError = lsa_last_error(this->Ctx);
        out.points.clear();
      }
    }
  }

  // SSKE.cxx:640-680: one float per input point, in scan order
  std::unordered_map<std::string, std::vector<float>> GetDebugArray() const
  {
    static const char* names[10] = {"sin_angle", "saliency", "depth_gap", "intensity_gap", "edge_keypoint", "plane_keypoint",
                                    "blob_keypoint", "edge_validity", "plane_validity", "blob_validity"};
    std::unordered_map<std::string, std::vector<float>> map;
    for (int i = 0; i < 10; ++i)
    {
      std::vector<float> v(this->ScanSize);
      if (this->ScanSize > 0) lsa_download_debug(this->Ctx, i, v.data(), this->ScanSize);
      map[names[i]] = std::move(v);
    }
    return map;
  }

  const std::string& GetLastError() const { return this->LastError; }
  lsa_ctx* GetContext() { return this->Ctx; }

private:
  lsa_ctx* Ctx = nullptr;
  lsa_extract_params_t Params;
  std::map<Keypoint, PointCloud::Ptr> Keypoints;
  int NbThreads = 1;
  int ScanSize = 0;
  std::string LastError;
};

#undef LSA_SSKE_PARAM

}  // namespace LidarSlam
