// ORACLE -- TEST INFRASTRUCTURE ONLY (see orc_math.hpp header).  PARITY UNPINNED.
//
// CPU restatement of slam_lib/src/KeypointsMatcher.cxx.
#include "orc_matcher.hpp"

namespace orc
{
namespace
{
// Utils::ComputeMeanAndPCA<Point,double> (Utilities.h:247-262)
void MeanAndPCA(const std::vector<Point>& cloud, const std::vector<int>& indices, V3d& mean, M3d& evecs, double evals[3])
{
  M3d cov;
  mean_and_cov<double>(
    (int)indices.size(), [&](int i, float& x, float& y, float& z) { const Point& p = cloud[indices[i]]; x = p.x; y = p.y; z = p.z; },
    mean, cov);
  eigen33<double>(cov, evecs, evals);
}
}  // namespace

// KeypointsMatcher.cxx:33-74
MatchingResults KeypointsMatcher::BuildMatchResiduals(const std::vector<Point>& currPoints, const KDTree& prevPoints, Keypoint type)
{
  MatchingResults res;
  res.Reset(currPoints.size());
  if (!currPoints.empty() && !prevPoints.Empty())
  {
    const int n = (int)currPoints.size();
    #pragma omp parallel for num_threads(Params.NbThreads) schedule(guided, 8)
    for (int i = 0; i < n; ++i)
    {
      MatchInfo m;
      switch (type)
      {
        case EDGE: m = BuildLineMatch(prevPoints, currPoints[i]); break;
        case PLANE: m = BuildPlaneMatch(prevPoints, currPoints[i]); break;
        case BLOB: m = BuildBlobMatch(prevPoints, currPoints[i]); break;
        default: m = {UNKOWN, 0., Residual()}; break;
      }
      res.Rejections[i] = m.Status;
      res.Weights[i] = m.Weight;
      res.Residuals[i] = m.Cost;
      #pragma omp atomic
      res.RejectionsHistogram[m.Status]++;
    }
  }
  return res;
}

// KeypointsMatcher.cxx:78-103 (Ceres >= 2 branch: ScaledLoss(Tukey(sat), weight))
Residual KeypointsMatcher::BuildResidual(const double A[9], const V3d& P, const V3d& X, double weight) const
{
  Residual r;
  r.valid = true;
  std::memcpy(r.A, A, sizeof(r.A));
  r.P[0] = P.x; r.P[1] = P.y; r.P[2] = P.z;
  r.X[0] = X.x; r.X[1] = X.y; r.X[2] = X.z;
  r.weight = weight;
  r.sat = Params.SaturationDistance;
  return r;
}

// KeypointsMatcher.cxx:106-187
KeypointsMatcher::MatchInfo KeypointsMatcher::BuildLineMatch(const KDTree& tree, const Point& p)
{
  if (Params.EdgeNbNeighbors < 2 || Params.EdgeMinNbNeighbors < 2)
    return {BAD_MODEL_PARAMETRIZATION, 0., Residual()};

  V3d basePoint = {(double)p.x, (double)p.y, (double)p.z};
  V3d worldPoint = iso_apply(PosePrior, basePoint);
  const double pos[3] = {worldPoint.x, worldPoint.y, worldPoint.z};

  std::vector<int> knnIndices;
  std::vector<float> knnSqDist;
  if (Params.SingleEdgePerRing)
    GetPerRingLineNeighbors(tree, pos, Params.EdgeNbNeighbors, knnIndices, knnSqDist);
  else
    GetRansacLineNeighbors(tree, pos, Params.EdgeNbNeighbors, Params.EdgeMaxModelError, knnIndices, knnSqDist);

  unsigned neighborhoodSize = knnIndices.size();
  if (neighborhoodSize < Params.EdgeMinNbNeighbors)
    return {NOT_ENOUGH_NEIGHBORS, 0., Residual()};
  if (knnSqDist.back() > Params.MaxNeighborsDistance * Params.MaxNeighborsDistance)
    return {NEIGHBORS_TOO_FAR, 0., Residual()};

  V3d mean;
  M3d evecs;
  double evals[3];
  MeanAndPCA(*tree.GetInputCloud(), knnIndices, mean, evecs, evals);
  const V3d n = col(evecs, 2);
  const double nv[3] = {n.x, n.y, n.z};
  double A[9];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) A[i * 3 + j] = (i == j ? 1. : 0.) - nv[i] * nv[j];
  if (!std::isfinite(A[0]))
    return {INVALID_NUMERICAL, 0., Residual()};
  double mse = evals[0] + evals[1];
  if (mse >= Params.EdgeMaxModelError * Params.EdgeMaxModelError)
    return {MSE_TOO_LARGE, 0., Residual()};
  double fitQualityCoeff = (mse <= 1e-6) ? 1. : 1. - std::sqrt(mse) / Params.EdgeMaxModelError;
  return {SUCCESS, fitQualityCoeff, BuildResidual(A, mean, basePoint, fitQualityCoeff)};
}

// KeypointsMatcher.cxx:190-273
KeypointsMatcher::MatchInfo KeypointsMatcher::BuildPlaneMatch(const KDTree& tree, const Point& p)
{
  if (Params.PlaneNbNeighbors < 3)
    return {BAD_MODEL_PARAMETRIZATION, 0., Residual()};

  V3d basePoint = {(double)p.x, (double)p.y, (double)p.z};
  V3d worldPoint = iso_apply(PosePrior, basePoint);
  const double pos[3] = {worldPoint.x, worldPoint.y, worldPoint.z};

  std::vector<int> knnIndices(Params.PlaneNbNeighbors);
  std::vector<float> knnSqDist(Params.PlaneNbNeighbors);
  unsigned neighborhoodSize = tree.KnnSearch(pos, Params.PlaneNbNeighbors, knnIndices.data(), knnSqDist.data());
  knnIndices.resize(neighborhoodSize);
  knnSqDist.resize(neighborhoodSize);
  if (neighborhoodSize < Params.PlaneNbNeighbors)
    return {NOT_ENOUGH_NEIGHBORS, 0., Residual()};
  if (knnSqDist.back() > Params.MaxNeighborsDistance * Params.MaxNeighborsDistance)
    return {NEIGHBORS_TOO_FAR, 0., Residual()};

  V3d mean;
  M3d evecs;
  double evals[3];
  MeanAndPCA(*tree.GetInputCloud(), knnIndices, mean, evecs, evals);
  if (evals[1] / evals[2] < Params.PlanarityThreshold)
    return {BAD_PCA_STRUCTURE, 0., Residual()};
  const V3d n = col(evecs, 0);
  const double nv[3] = {n.x, n.y, n.z};
  double A[9];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) A[i * 3 + j] = nv[i] * nv[j];
  if (!std::isfinite(A[0]))
    return {INVALID_NUMERICAL, 0., Residual()};
  double mse = evals[0];
  if (mse >= Params.PlaneMaxModelError * Params.PlaneMaxModelError)
    return {MSE_TOO_LARGE, 0., Residual()};
  double fitQualityCoeff = (mse <= 1e-6) ? 1. : 1. - std::sqrt(mse) / Params.PlaneMaxModelError;
  return {SUCCESS, fitQualityCoeff, BuildResidual(A, mean, basePoint, fitQualityCoeff)};
}

// KeypointsMatcher.cxx:276-346
KeypointsMatcher::MatchInfo KeypointsMatcher::BuildBlobMatch(const KDTree& tree, const Point& p)
{
  if (Params.BlobNbNeighbors < 4)
    return {BAD_MODEL_PARAMETRIZATION, 0., Residual()};

  V3d basePoint = {(double)p.x, (double)p.y, (double)p.z};
  V3d worldPoint = iso_apply(PosePrior, basePoint);
  const double pos[3] = {worldPoint.x, worldPoint.y, worldPoint.z};

  std::vector<int> knnIndices(Params.BlobNbNeighbors);
  std::vector<float> knnSqDist(Params.BlobNbNeighbors);
  unsigned neighborhoodSize = tree.KnnSearch(pos, Params.BlobNbNeighbors, knnIndices.data(), knnSqDist.data());
  knnIndices.resize(neighborhoodSize);
  knnSqDist.resize(neighborhoodSize);
  if (neighborhoodSize < Params.BlobNbNeighbors)
    return {NOT_ENOUGH_NEIGHBORS, 0., Residual()};
  if (knnSqDist.back() > Params.MaxNeighborsDistance * Params.MaxNeighborsDistance)
    return {NEIGHBORS_TOO_FAR, 0., Residual()};

  V3d mean;
  M3d evecs;
  double evals[3];
  MeanAndPCA(*tree.GetInputCloud(), knnIndices, mean, evecs, evals);
  if (evals[0] <= 0. || evals[1] <= 0.)
    return {BAD_PCA_STRUCTURE, 0., Residual()};
  const double d[3] = {1. / std::sqrt(evals[0]), 1. / std::sqrt(evals[1]), 1. / std::sqrt(evals[2])};
  double A[9];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j)
      A[i * 3 + j] = ((evecs(i, 0) * d[0]) * evecs(j, 0) + (evecs(i, 1) * d[1]) * evecs(j, 1)) + (evecs(i, 2) * d[2]) * evecs(j, 2);
  if (!std::isfinite(A[0]) || !std::isfinite(d[0] * d[1] * d[2]))
    return {INVALID_NUMERICAL, 0., Residual()};
  return {SUCCESS, 1.0, BuildResidual(A, mean, basePoint, 1.0)};
}

// KeypointsMatcher.cxx:349-405
void KeypointsMatcher::GetPerRingLineNeighbors(const KDTree& tree, const double pos[3], unsigned knearest,
                                               std::vector<int>& validKnnIndices, std::vector<float>& validKnnSqDist) const
{
  std::vector<int> knnIndices(knearest);
  std::vector<float> knnSqDist(knearest);
  unsigned neighborhoodSize = tree.KnnSearch(pos, knearest, knnIndices.data(), knnSqDist.data());
  if (neighborhoodSize == 0)
    return;
  const std::vector<Point>& pts = *tree.GetInputCloud();
  int closestLaserId = pts[knnIndices[0]].laser_id;
  int laserIdMin = std::numeric_limits<int>::max();
  int laserIdMax = std::numeric_limits<int>::min();
  for (unsigned k = 0; k < neighborhoodSize; ++k)
  {
    int scanLine = pts[knnIndices[k]].laser_id;
    laserIdMin = std::min(laserIdMin, scanLine);
    laserIdMax = std::max(laserIdMax, scanLine);
  }
  int nLasers = laserIdMax - laserIdMin + 1;
  std::vector<uint8_t> idAlreadyTook(nLasers, 0);
  idAlreadyTook[closestLaserId - laserIdMin] = 1;
  const int maxScanLineDiff = 4;
  for (int laserId = laserIdMin; laserId <= laserIdMax; ++laserId)
    if (std::abs(closestLaserId - laserId) > maxScanLineDiff)
      idAlreadyTook[laserId - laserIdMin] = 1;
  validKnnIndices.clear();
  validKnnSqDist.clear();
  for (unsigned k = 0; k < neighborhoodSize; ++k)
  {
    int scanLine = pts[knnIndices[k]].laser_id - laserIdMin;
    if (!idAlreadyTook[scanLine])
    {
      idAlreadyTook[scanLine] = 1;
      validKnnIndices.push_back(knnIndices[k]);
      validKnnSqDist.push_back(knnSqDist[k]);
    }
  }
}

// KeypointsMatcher.cxx:408-480
void KeypointsMatcher::GetRansacLineNeighbors(const KDTree& tree, const double pos[3], unsigned knearest, double maxDistInlier,
                                              std::vector<int>& validKnnIndices, std::vector<float>& validKnnSqDist) const
{
  std::vector<int> knnIndices(knearest);
  std::vector<float> knnSqDist(knearest);
  unsigned neighborhoodSize = tree.KnnSearch(pos, knearest, knnIndices.data(), knnSqDist.data());
  if (neighborhoodSize < 2)
    return;
  const std::vector<Point>& pts = *tree.GetInputCloud();
  const float squaredMaxDistInlier = maxDistInlier * maxDistInlier;
  const V3f P1 = xyz(pts[knnIndices[0]]);

  std::vector<std::vector<unsigned>> inliersList;
  inliersList.reserve(neighborhoodSize - 1);
  for (unsigned ptIndex = 1; ptIndex < neighborhoodSize; ++ptIndex)
  {
    const V3f P2 = xyz(pts[knnIndices[ptIndex]]);
    V3f dir = normalized(sub(P2, P1));
    std::vector<unsigned> inlierIndex;
    for (unsigned candidateIndex = 1; candidateIndex < neighborhoodSize; ++candidateIndex)
    {
      if (candidateIndex == ptIndex)
        inlierIndex.push_back(candidateIndex);
      else
      {
        const V3f Pcdt = xyz(pts[knnIndices[candidateIndex]]);
        if (sqnorm(cross(sub(Pcdt, P1), dir)) < squaredMaxDistInlier)
          inlierIndex.push_back(candidateIndex);
      }
    }
    inliersList.push_back(inlierIndex);
  }
  std::size_t maxInliers = 0;
  int indexMaxInliers = -1;
  for (unsigned k = 0; k < inliersList.size(); ++k)
    if (inliersList[k].size() > maxInliers)
    {
      maxInliers = inliersList[k].size();
      indexMaxInliers = k;
    }
  validKnnIndices.clear();
  validKnnSqDist.clear();
  validKnnIndices.push_back(knnIndices[0]);
  validKnnSqDist.push_back(knnSqDist[0]);
  for (unsigned inlier : inliersList[indexMaxInliers])
  {
    validKnnIndices.push_back(knnIndices[inlier]);
    validKnnSqDist.push_back(knnSqDist[inlier]);
  }
}

}  // namespace orc
