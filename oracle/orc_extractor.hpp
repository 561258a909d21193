// ORACLE -- TEST INFRASTRUCTURE ONLY (see orc_math.hpp header).  PARITY UNPINNED.
//
// CPU restatement of LidarSlam::SpinningSensorKeypointExtractor
//   slam_lib/src/SpinningSensorKeypointExtractor.cxx (all)
//   slam_lib/include/LidarSlam/SpinningSensorKeypointExtractor.h:119-157 (defaults)
#pragma once
#include <vector>
#include <array>
#include "orc_math.hpp"

namespace orc
{

enum Keypoint { EDGE = 0, PLANE = 1, BLOB = 2, nKeypointTypes = 3 };

struct ExtractorParams
{
  int NbThreads = 1;
  int NeighborWidth = 4;
  float MinDistanceToSensor = 1.5f;
  float MinBeamSurfaceAngle = 10.f;
  float PlaneSinAngleThreshold = 0.5f;
  float EdgeSinAngleThreshold = 0.86f;
  float DistToLineThreshold = 0.20f;
  float EdgeDepthGapThreshold = 0.15f;
  float EdgeSaliencyThreshold = 1.5f;
  float EdgeIntensityGapThreshold = 50.f;
};

class Extractor
{
public:
  ExtractorParams P;
  float AzimuthalResolution = 0.f;
  unsigned NbLaserRings = 0;

  // SSKE.cxx:118-136
  void ComputeKeyPoints(const std::vector<Point>& scan);

  std::vector<Point> Keypoints[3];

  // Per-ring arrays (ring-major), as the reference keeps them
  std::vector<std::vector<Point>> ScanLines;
  std::vector<std::vector<float>> Angles, DepthGap, Saliency, IntensityGap;
  std::vector<std::vector<uint8_t>> IsPointValid, Label;  // 3-bit flags, bit k = Keypoint k

  // GetDebugArray (SSKE.cxx:640-680) flattened to scan order. id:
  // 0 sin_angle 1 saliency 2 depth_gap 3 intensity_gap 4..6 {edge,plane,blob}_keypoint
  // 7..9 {edge,plane,blob}_validity
  std::vector<float> DebugArray(int id) const;

private:
  const std::vector<Point>* Scan = nullptr;
  void ConvertAndSortScanLines();
  void PrepareDataForNextFrame();
  void InvalidateNotUsablePoints();
  void ComputeCurvature();
  void SetKeyPointsLabels();
  void EstimateAzimuthalResolution();
  bool IsScanLineAlmostEmpty(int n) const { return n < 2 * P.NeighborWidth + 1; }
};

}  // namespace orc
