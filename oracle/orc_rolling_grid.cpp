// ORACLE -- TEST INFRASTRUCTURE ONLY (see orc_math.hpp header).  PARITY UNPINNED.
//
// CPU restatement of slam_lib/src/RollingGrid.cxx.  Mixed float/double
// expressions follow Eigen's scalar promotion: an Array3f combined with a
// double scalar converts the scalar to float first.
#include "orc_rolling_grid.hpp"
#include <cmath>

namespace orc
{
namespace
{
// Utils::PositionToVoxel (RollingGrid.h:39-42)
inline void PositionToVoxel(const float p[3], const float origin[3], double resolution, int out[3])
{
  const float r = (float)resolution;
  for (int i = 0; i < 3; ++i)
  {
    // the reference casts whatever it gets; out of range (the +-FLT_MAX box of an empty cloud) that is
    // undefined in C++ and INT_MIN on x86-64, which is stated here explicitly
    const float v = std::round((p[i] - origin[i]) / r);
    out[i] = (v >= -2147483648.f && v < 2147483648.f) ? (int)v : std::numeric_limits<int>::min();
  }
}
inline void MinMax3D(const std::vector<Point>& c, float mn[3], float mx[3])
{
  for (int i = 0; i < 3; ++i) { mn[i] = std::numeric_limits<float>::max(); mx[i] = -std::numeric_limits<float>::max(); }
  for (const Point& p : c)
  {
    const float v[3] = {p.x, p.y, p.z};
    for (int i = 0; i < 3; ++i) { mn[i] = std::min(mn[i], v[i]); mx[i] = std::max(mx[i], v[i]); }
  }
}
}  // namespace

// RollingGrid.cxx:40-48
void RollingGrid::Reset(const float position[3])
{
  Clear();
  const float r = (float)VoxelResolution;
  for (int i = 0; i < 3; ++i)
    VoxelGridPosition[i] = std::floor((position ? position[i] : 0.f) / r) * r;
}

// :51-56
void RollingGrid::Clear()
{
  NbPoints = 0;
  Voxels.clear();
  KdTree.Reset(nullptr);
}

// :59-70
void RollingGrid::SetGridSize(int size)
{
  GridSize = size;
  std::vector<Point> prevMap = Get();
  Clear();
  if (!prevMap.empty())
    Add(prevMap);
}

// :73-88
void RollingGrid::SetVoxelResolution(double resolution)
{
  VoxelResolution = int(resolution / LeafSize) * LeafSize;
  const float r = (float)VoxelResolution;
  for (int i = 0; i < 3; ++i) VoxelGridPosition[i] = std::floor(VoxelGridPosition[i] / r) * r;
  std::vector<Point> prevMap = Get();
  Clear();
  if (!prevMap.empty())
    Add(prevMap);
}

// :95-114
std::vector<Point> RollingGrid::Get(bool clean) const
{
  std::vector<Point> pc;
  pc.reserve(NbPoints);
  ForEachVoxel([&](int, const Voxel& v) {
    if (!clean || v.count > MinFramesPerVoxel) pc.push_back(v.point);
  });
  return pc;
}

// :117-157
void RollingGrid::Roll(const float minPoint[3], const float maxPoint[3])
{
  const double halfGridSize = static_cast<double>(GridSize) / 2 * VoxelResolution;
  const float h = (float)halfGridSize;
  const float r = (float)VoxelResolution;
  int voxelsOffset[3];
  bool any = false;
  for (int i = 0; i < 3; ++i)
  {
    float down = minPoint[i] - (VoxelGridPosition[i] - h);
    float up = maxPoint[i] - (VoxelGridPosition[i] + h);
    float off = (up + down) / 2.f;
    off = std::min(std::max(off, std::min(down, 0.f)), std::max(up, 0.f));
    voxelsOffset[i] = (int)std::round(off / r);
    any |= voxelsOffset[i] != 0;
  }
  if (!any)
    return;
  unsigned newNbPoints = 0;
  RollingVG newVoxels;
  for (auto& kvOut : Voxels)
  {
    int idx3d[3];
    To3d(kvOut.first, idx3d);
    bool in = true;
    for (int i = 0; i < 3; ++i)
    {
      idx3d[i] -= voxelsOffset[i];
      in &= (0 <= idx3d[i]) && (idx3d[i] < GridSize);
    }
    if (in)
    {
      newNbPoints += kvOut.second.size();
      newVoxels[To1d(idx3d)] = std::move(kvOut.second);
    }
  }
  NbPoints = newNbPoints;
  Voxels.swap(newVoxels);
  for (int i = 0; i < 3; ++i) VoxelGridPosition[i] += (float)voxelsOffset[i] * r;
}

// :160-318
void RollingGrid::Add(const std::vector<Point>& pointcloud, bool fixed, double currentTime, bool roll)
{
  if (pointcloud.empty())
    return;
  if (roll)
  {
    float mn[3], mx[3];
    MinMax3D(pointcloud, mn, mx);
    Roll(mn, mx);
  }
  const float res = (float)VoxelResolution;
  float voxelGridOrigin[3];
  for (int i = 0; i < 3; ++i) voxelGridOrigin[i] = VoxelGridPosition[i] - (float)(int(GridSize / 2) * VoxelResolution);

  std::unordered_map<int, std::unordered_map<int, bool>> seen;
  std::unordered_map<int, std::unordered_map<int, Voxel>> meanPts;
  bool updated = false;
  for (const Point& point : pointcloud)
  {
    const float p[3] = {point.x, point.y, point.z};
    int voxelCoordOut[3];
    PositionToVoxel(p, voxelGridOrigin, VoxelResolution, voxelCoordOut);
    bool in = true;
    for (int i = 0; i < 3; ++i) in &= (0 <= voxelCoordOut[i]) && (voxelCoordOut[i] < GridSize);
    if (!in)
      continue;
    float voxelGridCenterIn[3];
    for (int i = 0; i < 3; ++i) voxelGridCenterIn[i] = (float)voxelCoordOut[i] * res + voxelGridOrigin[i];
    int voxelCoordIn[3];
    PositionToVoxel(p, voxelGridCenterIn, LeafSize, voxelCoordIn);
    unsigned idxOut = To1d(voxelCoordOut);
    unsigned idxIn = To1d(voxelCoordIn);
    if (!Voxels.count(idxOut) || !Voxels[idxOut].count(idxIn))
    {
      Voxels[idxOut][idxIn].point = point;
      ++NbPoints;
      updated = true;
    }
    else
    {
      auto& voxel = Voxels[idxOut][idxIn];
      if (voxel.point.label == 1)
        continue;
      switch (Sampling)
      {
        case SamplingMode::FIRST:
          break;
        case SamplingMode::LAST:
          voxel.point = point;
          updated = true;
          break;
        case SamplingMode::MAX_INTENSITY:
          if (point.intensity > voxel.point.intensity)
          {
            voxel.point = point;
            updated = true;
          }
          break;
        case SamplingMode::CENTER_POINT:
        {
          V3f c;
          float* cc[3] = {&c.x, &c.y, &c.z};
          for (int i = 0; i < 3; ++i)
            *cc[i] = voxelGridCenterIn[i] - res / 2.f + (float)LeafSize * (float)voxelCoordIn[i];
          if (norm(sub(xyz(point), c)) < norm(sub(xyz(voxel.point), c)))
          {
            voxel.point = point;
            updated = true;
          }
          break;
        }
        case SamplingMode::CENTROID:
        {
          Voxel& v = meanPts[idxOut][idxIn];
          const float cnt = (float)v.count;
          v.point.x = (v.point.x * cnt + point.x) / (float)(v.count + 1);
          v.point.y = (v.point.y * cnt + point.y) / (float)(v.count + 1);
          v.point.z = (v.point.z * cnt + point.z) / (float)(v.count + 1);
          ++v.count;
          break;
        }
      }
    }
    // RollingGrid.cxx:282-297 (this block sits inside the per-point loop in the reference)
    if (Sampling == SamplingMode::CENTROID)
    {
      for (auto& vOut : meanPts)
        for (auto& vIn : vOut.second)
        {
          auto& voxel = Voxels[vOut.first][vIn.first];
          const float cnt = (float)voxel.count;
          voxel.point.x = (voxel.point.x * cnt + vIn.second.point.x) / (float)(voxel.count + 1);
          voxel.point.y = (voxel.point.y * cnt + vIn.second.point.y) / (float)(voxel.count + 1);
          voxel.point.z = (voxel.point.z * cnt + vIn.second.point.z) / (float)(voxel.count + 1);
        }
    }
    auto& voxel = Voxels[idxOut][idxIn];
    voxel.point.time = currentTime;
    voxel.point.label = fixed ? 1 : 0;
    if (!seen.count(idxOut) || !seen[idxOut].count(idxIn))
    {
      ++voxel.count;
      seen[idxOut][idxIn] = true;
    }
  }
  if (updated)
    KdTree.Reset(nullptr);
}

// :325-351
void RollingGrid::ClearOldPoints(double currentTime)
{
  auto itOut = Voxels.begin();
  while (itOut != Voxels.end())
  {
    auto itIn = itOut->second.begin();
    while (itIn != itOut->second.end())
    {
      Voxel& voxel = itIn->second;
      if (!voxel.point.label && currentTime - voxel.point.time > DecayingThreshold)
        itIn = itOut->second.erase(itIn);
      else
        ++itIn;
    }
    if (itOut->second.empty())
      itOut = Voxels.erase(itOut);
    else
      ++itOut;
  }
}

// :354-360
void RollingGrid::BuildSubMapKdTree()
{
  SubMap = Get();
  KdTree.Reset(&SubMap);
}

// :363-442
void RollingGrid::BuildSubMapKdTree(const float minPoint[3], const float maxPoint[3], int minNbPoints)
{
  float voxelGridOrigin[3];
  for (int i = 0; i < 3; ++i) voxelGridOrigin[i] = VoxelGridPosition[i] - (float)(int(GridSize / 2) * VoxelResolution);
  int imin[3], imax[3];
  PositionToVoxel(minPoint, voxelGridOrigin, VoxelResolution, imin);
  PositionToVoxel(maxPoint, voxelGridOrigin, VoxelResolution, imax);
  for (int i = 0; i < 3; ++i) { imin[i] = std::max(imin[i], 0); imax[i] = std::min(imax[i], GridSize - 1); }
  auto inside = [&](int id) {
    int v[3];
    To3d(id, v);
    return imin[0] <= v[0] && v[0] <= imax[0] && imin[1] <= v[1] && v[1] <= imax[1] && imin[2] <= v[2] && v[2] <= imax[2];
  };
  SubMap.clear();
  SubMap.reserve(NbPoints);
  if (minNbPoints < 0 || MinFramesPerVoxel <= 1)
  {
    ForEachVoxel([&](int idxOut, const Voxel& v) {
      if (inside(idxOut)) SubMap.push_back(v.point);
    });
  }
  else
  {
    ForEachVoxel([&](int idxOut, const Voxel& v) {
      if (inside(idxOut) && (v.count >= MinFramesPerVoxel || v.point.label == 1)) SubMap.push_back(v.point);
    });
    if (int(SubMap.size()) < minNbPoints)
    {
      ForEachVoxel([&](int idxOut, const Voxel& v) {
        if (inside(idxOut) && v.count < MinFramesPerVoxel && v.point.label != 1) SubMap.push_back(v.point);
      });
    }
  }
  KdTree.Reset(&SubMap);
}

}  // namespace orc
