// ORACLE -- TEST INFRASTRUCTURE ONLY (see orc_math.hpp header).  PARITY UNPINNED.
//
// CPU restatement of LidarSlam::KeypointsMatcher
//   slam_lib/src/KeypointsMatcher.cxx (all), slam_lib/include/LidarSlam/KeypointsMatcher.h:43-122
#pragma once
#include <vector>
#include <array>
#include "orc_math.hpp"
#include "orc_kdtree.hpp"
#include "orc_extractor.hpp"

namespace orc
{

// KeypointsMatcher::Parameters (KeypointsMatcher.h:43-77)
struct MatchParams
{
  unsigned NbThreads = 1;
  bool SingleEdgePerRing = false;
  double MaxNeighborsDistance = 5.;
  unsigned EdgeNbNeighbors = 10;
  unsigned EdgeMinNbNeighbors = 4;
  double EdgeMaxModelError = 0.2;
  unsigned PlaneNbNeighbors = 5;
  double PlanarityThreshold = 0.04;
  double PlaneMaxModelError = 0.2;
  unsigned BlobNbNeighbors = 10;
  double SaturationDistance = 1.;
};

// MatchingResults::MatchStatus (KeypointsMatcher.h:82-93)
enum MatchStatus : uint8_t
{
  SUCCESS = 0,
  BAD_MODEL_PARAMETRIZATION,
  NOT_ENOUGH_NEIGHBORS,
  NEIGHBORS_TOO_FAR,
  BAD_PCA_STRUCTURE,
  INVALID_NUMERICAL,
  MSE_TOO_LARGE,
  UNKOWN,
  nStatus
};

// What CeresTools::Residual carries for the point-to-model cost
// (MahalanobisDistanceAffineIsometryResidual + ScaledLoss(TukeyLoss(sat), weight),
//  KeypointsMatcher.cxx:78-103, CeresCostFunctions.h:105-152)
struct Residual
{
  bool valid = false;
  double A[9];  // row-major
  double P[3];
  double X[3];
  double weight = 0.;
  double sat = 1.;  // Tukey scale a = SaturationDistance
};

struct MatchingResults
{
  std::vector<Residual> Residuals;
  std::vector<uint8_t> Rejections;
  std::vector<double> Weights;
  std::array<int, nStatus> RejectionsHistogram{};
  unsigned NbMatches() const { return RejectionsHistogram[SUCCESS]; }
  void Reset(unsigned N)
  {
    Weights.assign(N, 0.);
    Rejections.assign(N, UNKOWN);
    RejectionsHistogram.fill(0);
    Residuals.assign(N, Residual());
  }
};

class KeypointsMatcher
{
public:
  KeypointsMatcher(const MatchParams& params, const Iso& posePrior) : Params(params), PosePrior(posePrior) {}
  MatchingResults BuildMatchResiduals(const std::vector<Point>& currPoints, const KDTree& prevPoints, Keypoint type);

private:
  struct MatchInfo { MatchStatus Status; double Weight; Residual Cost; };
  MatchInfo BuildLineMatch(const KDTree& tree, const Point& p);
  MatchInfo BuildPlaneMatch(const KDTree& tree, const Point& p);
  MatchInfo BuildBlobMatch(const KDTree& tree, const Point& p);
  void GetPerRingLineNeighbors(const KDTree& tree, const double pos[3], unsigned knearest, std::vector<int>& idx, std::vector<float>& d2) const;
  void GetRansacLineNeighbors(const KDTree& tree, const double pos[3], unsigned knearest, double maxDistInlier, std::vector<int>& idx, std::vector<float>& d2) const;
  Residual BuildResidual(const double A[9], const V3d& P, const V3d& X, double weight) const;

  const MatchParams Params;
  const Iso PosePrior;
};

}  // namespace orc
