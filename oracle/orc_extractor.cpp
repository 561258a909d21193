// ORACLE -- TEST INFRASTRUCTURE ONLY (see orc_math.hpp header).  PARITY UNPINNED.
//
// CPU restatement of slam_lib/src/SpinningSensorKeypointExtractor.cxx.
// Deliberate, documented deviations from the reference:
//  * Utils::SortIdx uses an unstable std::sort on the score alone
//    (slam_lib/include/LidarSlam/Utilities.h:96-110); here ties are broken by
//    ascending index so that the order is a total one and the GPU can match it.
//  * NaN scores never become keypoints (the reference's behaviour on NaN is
//    whatever std::sort leaves, i.e. undefined).
#include "orc_extractor.hpp"
#include <cmath>
#include <numeric>
#include <cstdio>

namespace orc
{
namespace
{
// SSKE.cxx:33-115
struct LineFitting
{
  V3f Direction{0, 0, 0};
  V3f Position{0, 0, 0};
  float MaxDistance = 0.02f;
  float MaxAngle = float(40. * 0.017453293);  // PCL DEG2RAD macro

  float SquaredDistanceToPoint(const V3f& p) const { return sqnorm(cross(sub(p, Position), Direction)); }

  // SSKE.cxx:59-84
  bool FitPCA(const std::vector<Point>& cloud, const int* indices, int n)
  {
    M3<float> cov, evecs;
    float evals[3];
    mean_and_cov<float>(
      n, [&](int i, float& x, float& y, float& z) { const Point& p = cloud[indices[i]]; x = p.x; y = p.y; z = p.z; },
      Position, cov);
    eigen33<float>(cov, evecs, evals);
    Direction = col(evecs, 2);
    const float sqMaxDistance = MaxDistance * MaxDistance;
    for (int i = 0; i < n; ++i)
      if (SquaredDistanceToPoint(xyz(cloud[indices[i]])) > sqMaxDistance)
        return false;
    return true;
  }

  // SSKE.cxx:87-108
  bool FitPCAAndCheckConsistency(const std::vector<Point>& cloud, const int* indices, int n)
  {
    const float maxSinAngle = std::sin(MaxAngle);
    const V3f U = normalized(sub(xyz(cloud[indices[n - 1]]), xyz(cloud[indices[0]])));
    for (int i = 0; i + 1 < n; ++i)
    {
      const V3f V = normalized(sub(xyz(cloud[indices[i + 1]]), xyz(cloud[indices[i]])));
      const float sinAngle = norm(cross(U, V));
      if (sinAngle > maxSinAngle)
        return false;
    }
    return FitPCA(cloud, indices, n);
  }
};

inline float deg2rad_f(float deg) { return float(deg / 180. * M_PI); }
}  // namespace

// SSKE.cxx:118-136
void Extractor::ComputeKeyPoints(const std::vector<Point>& scan)
{
  Scan = &scan;
  ConvertAndSortScanLines();
  PrepareDataForNextFrame();
  InvalidateNotUsablePoints();
  ComputeCurvature();
  SetKeyPointsLabels();
}

// SSKE.cxx:139-171
void Extractor::ConvertAndSortScanLines()
{
  for (auto& l : ScanLines) l.clear();
  for (const Point& p : *Scan)
  {
    while (p.laser_id >= ScanLines.size()) ScanLines.emplace_back();
    ScanLines[p.laser_id].push_back(p);
  }
  NbLaserRings = ScanLines.size();
  if (AzimuthalResolution < 1e-6 || M_PI / 4. < AzimuthalResolution)
    EstimateAzimuthalResolution();
}

// SSKE.cxx:174-204
void Extractor::PrepareDataForNextFrame()
{
  for (int k = 0; k < 3; ++k) Keypoints[k].clear();
  Angles.resize(NbLaserRings); Saliency.resize(NbLaserRings); DepthGap.resize(NbLaserRings);
  IntensityGap.resize(NbLaserRings); IsPointValid.resize(NbLaserRings); Label.resize(NbLaserRings);
  for (unsigned r = 0; r < NbLaserRings; ++r)
  {
    size_t n = ScanLines[r].size();
    IsPointValid[r].assign(n, 7);
    Label[r].assign(n, 0);
    Angles[r].assign(n, 0.f); Saliency[r].assign(n, 0.f); DepthGap[r].assign(n, 0.f); IntensityGap[r].assign(n, 0.f);
  }
}

// SSKE.cxx:207-308
void Extractor::InvalidateNotUsablePoints()
{
  const float angleBeamNormal = deg2rad_f(90 - P.MinBeamSurfaceAngle);
  float azimuthalResolution = AzimuthalResolution;
  if (azimuthalResolution < 1e-6 || M_PI / 4 < azimuthalResolution)
    azimuthalResolution = float(0.2 / 180. * M_PI);
  const float maxPosDiffCoeff = std::sin(azimuthalResolution) / std::cos(azimuthalResolution + angleBeamNormal);
  const int W = P.NeighborWidth;

  #pragma omp parallel for num_threads(P.NbThreads) schedule(guided)
  for (int scanLine = 0; scanLine < (int)NbLaserRings; ++scanLine)
  {
    const std::vector<Point>& cloud = ScanLines[scanLine];
    std::vector<uint8_t>& valid = IsPointValid[scanLine];
    const int Npts = cloud.size();
    if (IsScanLineAlmostEmpty(Npts))
    {
      for (int i = 0; i < Npts; ++i) valid[i] = 0;
      continue;
    }
    for (int i = 0; i < W; ++i)
    {
      valid[i] = 0;
      valid[Npts - 1 - i] = 0;
    }
    for (int index = W; index < Npts - W; ++index)
    {
      const V3f cur = xyz(cloud[index]);
      const float L = norm(cur);
      if (L < P.MinDistanceToSensor)
        valid[index] = 0;
      const float maxPosDiff = std::max(L * maxPosDiffCoeff, 0.02f);
      const float sqMaxPosDiff = maxPosDiff * maxPosDiff;
      const V3f nxt = xyz(cloud[index + 1]);
      if (sqnorm(sub(nxt, cur)) > sqMaxPosDiff)
      {
        if (L < norm(nxt))
        {
          valid[index + 1] = 0;
          for (int i = index + 1; i < index + W; ++i)
          {
            if (sqnorm(sub(xyz(cloud[i + 1]), xyz(cloud[i]))) > sqMaxPosDiff)
              break;
            valid[i + 1] = 0;
          }
        }
        else
        {
          valid[index] = 0;
          for (int i = index - 1; i > index - W; --i)
          {
            if (sqnorm(sub(xyz(cloud[i + 1]), xyz(cloud[i]))) > sqMaxPosDiff)
              break;
            valid[i] = 0;
          }
        }
      }
    }
  }
}

// SSKE.cxx:311-471
void Extractor::ComputeCurvature()
{
  const float sqDistToLineThreshold = P.DistToLineThreshold * P.DistToLineThreshold;
  const float sqDepthDistCoeff = 0.25f;
  const float minDepthGapDist = 1.5f;
  const int W = P.NeighborWidth;

  #pragma omp parallel for num_threads(P.NbThreads) schedule(guided)
  for (int scanLine = 0; scanLine < (int)NbLaserRings; ++scanLine)
  {
    const std::vector<Point>& cloud = ScanLines[scanLine];
    const int Npts = cloud.size();
    if (IsScanLineAlmostEmpty(Npts))
      continue;
    std::vector<int> leftNeighbors(W), rightNeighbors(W), farNeighbors;
    farNeighbors.reserve(2 * W);

    for (int index = W; (index + W) < Npts; ++index)
    {
      if (IsPointValid[scanLine][index] == 0)
        continue;
      const V3f centralPoint = xyz(cloud[index]);
      IntensityGap[scanLine][index] = std::abs(cloud[index + 1].intensity - cloud[index - 1].intensity);

      LineFitting leftLine, rightLine;
      for (int j = index - 1; j >= index - W; --j) leftNeighbors[index - 1 - j] = j;
      for (int j = index + 1; j <= index + W; ++j) rightNeighbors[j - index - 1] = j;

      const bool leftFlat = leftLine.FitPCAAndCheckConsistency(cloud, leftNeighbors.data(), W);
      const bool rightFlat = rightLine.FitPCAAndCheckConsistency(cloud, rightNeighbors.data(), W);

      float distLeft = 0.f, distRight = 0.f;
      if (leftFlat && rightFlat)
      {
        distLeft = leftLine.SquaredDistanceToPoint(centralPoint);
        distRight = rightLine.SquaredDistanceToPoint(centralPoint);
        if ((distLeft < sqDistToLineThreshold) && (distRight < sqDistToLineThreshold))
          Angles[scanLine][index] = norm(cross(leftLine.Direction, rightLine.Direction));
      }
      else if (!leftFlat && rightFlat)
      {
        distLeft = std::numeric_limits<float>::max();
        for (int id : leftNeighbors)
          distLeft = std::min(distLeft, rightLine.SquaredDistanceToPoint(xyz(cloud[id])));
        distLeft *= sqDepthDistCoeff;
      }
      else if (leftFlat && !rightFlat)
      {
        distRight = std::numeric_limits<float>::max();
        for (int id : rightNeighbors)
          distRight = std::min(distRight, leftLine.SquaredDistanceToPoint(xyz(cloud[id])));
        distRight *= sqDepthDistCoeff;
      }
      else
      {
        const float sqCurrDepth = sqnorm(centralPoint);
        bool hasLeft = false, hasRight = false;
        farNeighbors.clear();
        for (int id : leftNeighbors)
        {
          if (std::abs(sqnorm(xyz(cloud[id])) - sqCurrDepth) > minDepthGapDist)
          {
            hasLeft = true;
            farNeighbors.push_back(id);
          }
          else if (hasLeft)
            break;
        }
        for (int id : rightNeighbors)
        {
          if (std::abs(sqnorm(xyz(cloud[id])) - sqCurrDepth) > minDepthGapDist)
          {
            hasRight = true;
            farNeighbors.push_back(id);
          }
          else if (hasRight)
            break;
        }
        if (farNeighbors.size() > static_cast<unsigned>(W))
        {
          LineFitting farLine;
          farLine.FitPCA(cloud, farNeighbors.data(), farNeighbors.size());
          Saliency[scanLine][index] = farLine.SquaredDistanceToPoint(centralPoint);
        }
      }
      DepthGap[scanLine][index] = std::max(distLeft, distRight);
    }
  }
}

namespace
{
// Utils::SortIdx(v, false) with a total order: value descending, index
// ascending; NaN sorts last.
std::vector<int> SortIdxDesc(const std::vector<float>& v)
{
  std::vector<int> idx(v.size());
  std::iota(idx.begin(), idx.end(), 0);
  auto key = [&v](int i) { return std::isnan(v[i]) ? -std::numeric_limits<float>::infinity() : v[i]; };
  std::sort(idx.begin(), idx.end(), [&](int a, int b) {
    float ka = key(a), kb = key(b);
    return ka > kb || (ka == kb && a < b);
  });
  return idx;
}
}  // namespace

// SSKE.cxx:474-590
void Extractor::SetKeyPointsLabels()
{
  const float sqEdgeSaliencythreshold = P.EdgeSaliencyThreshold * P.EdgeSaliencyThreshold;
  const float sqEdgeDepthGapThreshold = P.EdgeDepthGapThreshold * P.EdgeDepthGapThreshold;

  #pragma omp parallel for num_threads(P.NbThreads) schedule(guided)
  for (int scanLine = 0; scanLine < (int)NbLaserRings; ++scanLine)
  {
    const int Npts = ScanLines[scanLine].size();
    if (IsScanLineAlmostEmpty(Npts))
      continue;
    std::vector<uint8_t>& valid = IsPointValid[scanLine];
    std::vector<uint8_t>& label = Label[scanLine];

    std::vector<int> sortedDepthGapIdx = SortIdxDesc(DepthGap[scanLine]);
    std::vector<int> sortedAnglesIdx = SortIdxDesc(Angles[scanLine]);
    std::vector<int> sortedSaliencyIdx = SortIdxDesc(Saliency[scanLine]);
    std::vector<int> sortedIntensityGap = SortIdxDesc(IntensityGap[scanLine]);

    auto addEdgesUsingCriterion = [&](const std::vector<int>& sorted, const std::vector<float>& values, float threshold,
                                      int invalidNeighborhoodSize) {
      for (int index : sorted)
      {
        if (!(values[index] >= threshold))  // reference: `values < threshold -> break` (NaN also stops here)
          break;
        if (!(valid[index] & (1 << EDGE)))
          continue;
        label[index] |= (1 << EDGE);
        const int indexBegin = std::max(0, index - invalidNeighborhoodSize);
        const int indexEnd = std::min(Npts - 1, index + invalidNeighborhoodSize);
        for (int j = indexBegin; j <= indexEnd; ++j)
          valid[j] &= ~(1 << EDGE);
      }
    };
    addEdgesUsingCriterion(sortedDepthGapIdx, DepthGap[scanLine], sqEdgeDepthGapThreshold, P.NeighborWidth - 1);
    addEdgesUsingCriterion(sortedAnglesIdx, Angles[scanLine], P.EdgeSinAngleThreshold, P.NeighborWidth);
    addEdgesUsingCriterion(sortedSaliencyIdx, Saliency[scanLine], sqEdgeSaliencythreshold, P.NeighborWidth - 1);
    addEdgesUsingCriterion(sortedIntensityGap, IntensityGap[scanLine], P.EdgeIntensityGapThreshold, 1);

    // Planes (SSKE.cxx:536-563)
    for (int k = Npts - 1; k >= 0; --k)
    {
      int index = sortedAnglesIdx[k];
      const float sinAngle = Angles[scanLine][index];
      if (std::isnan(sinAngle))
        continue;
      if (sinAngle > P.PlaneSinAngleThreshold)
        break;
      if (!(valid[index] & (1 << PLANE)) || sinAngle < 1e-6)
        continue;
      label[index] |= (1 << PLANE);
      const int indexBegin = std::max(0, index - 4);
      const int indexEnd = std::min(Npts - 1, index + 4);
      for (int j = indexBegin; j <= indexEnd; ++j)
        valid[j] &= ~(1 << PLANE);
    }

    // Blobs (SSKE.cxx:568-572)
    for (int index = 0; index < Npts; index += 3)
      if (valid[index] & (1 << BLOB))
        label[index] |= (1 << BLOB);
  }

  // SSKE.cxx:575-589
  for (unsigned scanLine = 0; scanLine < NbLaserRings; ++scanLine)
  {
    const std::vector<Point>& cloud = ScanLines[scanLine];
    for (unsigned index = 0; index < cloud.size(); ++index)
      for (int k = 0; k < 3; ++k)
        if (Label[scanLine][index] & (1 << k))
        {
          IsPointValid[scanLine][index] |= (1 << k);
          Keypoints[k].push_back(cloud[index]);
        }
  }
}

// SSKE.cxx:593-637
void Extractor::EstimateAzimuthalResolution()
{
  std::vector<float> angles;
  angles.reserve(Scan->size());
  for (const auto& line : ScanLines)
  {
    for (unsigned index = 1; index < line.size(); ++index)
    {
      const Point& a = line[index - 1];
      const Point& b = line[index];
      float d = a.x * b.x + a.y * b.y;
      float na = std::sqrt(a.x * a.x + a.y * a.y), nb = std::sqrt(b.x * b.x + b.y * b.y);
      float angle = std::abs(std::acos(d / (na * nb)));
      if (angle > 1e-4)
        angles.push_back(angle);
    }
  }
  if (angles.size() < 100)
    return;
  std::sort(angles.begin(), angles.end());
  unsigned maxInliersIdx = angles.size();
  float maxAngle = float(5. / 180. * M_PI);
  float medianAngle = 0.f;
  while (maxAngle > 1.8 * medianAngle)
  {
    maxInliersIdx = std::upper_bound(angles.begin(), angles.begin() + maxInliersIdx, maxAngle) - angles.begin();
    medianAngle = angles[maxInliersIdx / 2];
    maxAngle = std::min(medianAngle * 2., maxAngle / 1.8);
  }
  AzimuthalResolution = medianAngle;
}

// SSKE.cxx:640-680
std::vector<float> Extractor::DebugArray(int id) const
{
  std::vector<float> v(Scan->size());
  std::vector<int> indexByScanLine(NbLaserRings, 0);
  for (size_t i = 0; i < Scan->size(); ++i)
  {
    const unsigned r = (*Scan)[i].laser_id;
    const int j = indexByScanLine[r]++;
    switch (id)
    {
      case 0: v[i] = Angles[r][j]; break;
      case 1: v[i] = Saliency[r][j]; break;
      case 2: v[i] = DepthGap[r][j]; break;
      case 3: v[i] = IntensityGap[r][j]; break;
      case 4: case 5: case 6: v[i] = (Label[r][j] >> (id - 4)) & 1; break;
      default: v[i] = (IsPointValid[r][j] >> (id - 7)) & 1; break;
    }
  }
  return v;
}

}  // namespace orc
