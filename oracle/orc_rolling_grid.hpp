// ORACLE -- TEST INFRASTRUCTURE ONLY (see orc_math.hpp header).  PARITY UNPINNED.
//
// CPU restatement of LidarSlam::RollingGrid (target construction for the ICP)
//   slam_lib/src/RollingGrid.cxx:40-48, 73-88, 117-318, 353-463
//   slam_lib/include/LidarSlam/RollingGrid.h:39-42, 170-212
// The containers are the same std::unordered_map<int, ...> as the reference, so
// that the iteration order -- which fixes the order of the sub-map points and
// therefore the summation order of every PCA fed from it -- is libstdc++'s,
// exactly as in a reference build on this platform.
#pragma once
#include <algorithm>
#include <unordered_map>
#include <vector>
#include "orc_math.hpp"
#include "orc_kdtree.hpp"

namespace orc
{

enum class SamplingMode { FIRST = 0, LAST = 1, MAX_INTENSITY = 2, CENTER_POINT = 3, CENTROID = 4 };

class RollingGrid
{
public:
  struct Voxel { Point point{}; unsigned count = 0; };
  using SamplingVG = std::unordered_map<int, Voxel>;
  using RollingVG = std::unordered_map<int, SamplingVG>;

  RollingGrid() { Reset(); }
  void Reset(const float position[3] = nullptr);
  void Clear();
  void SetGridSize(int size);
  void SetVoxelResolution(double resolution);
  void SetLeafSize(double s) { LeafSize = s; }
  double GetLeafSize() const { return LeafSize; }
  void SetMinFramesPerVoxel(unsigned n) { MinFramesPerVoxel = n; }
  void SetSampling(SamplingMode m) { Sampling = m; }
  void SetDecayingThreshold(double d) { DecayingThreshold = d; }
  // Order in which Get / BuildSubMapKdTree hand the voxels out.  false: the iteration order of the reference's own
  // containers (std::unordered_map of libstdc++: an accident of the hash tables' history).  true (default): ascending
  // (outer voxel index, leaf voxel index as unsigned) -- the DEFINED order the device map uses; the product and the
  // oracle adopt it together, the reference's order stays available for comparison.
  void SetOrdered(bool b) { Ordered = b; }
  bool GetOrdered() const { return Ordered; }
  bool IsTimeThreshold() const { return DecayingThreshold > 0; }

  std::vector<Point> Get(bool clean = false) const;
  unsigned Size() const { return NbPoints; }
  void Roll(const float minPoint[3], const float maxPoint[3]);
  void Add(const std::vector<Point>& pointcloud, bool fixed = false, double currentTime = -1., bool roll = true);
  void BuildSubMapKdTree();
  void BuildSubMapKdTree(const float minPoint[3], const float maxPoint[3], int minNbPoints = -1);
  bool IsSubMapKdTreeValid() const { return !KdTree.Empty(); }
  const KDTree& GetSubMapKdTree() const { return KdTree; }
  const std::vector<Point>& GetSubMap() const { return SubMap; }
  void ClearOldPoints(double currentTime);

private:
  int GridSize = 50;
  double VoxelResolution = 10.;
  double LeafSize = 0.2;
  RollingVG Voxels;
  float VoxelGridPosition[3] = {0, 0, 0};
  unsigned NbPoints = 0;
  KDTree KdTree;
  std::vector<Point> SubMap;
  unsigned MinFramesPerVoxel = 0;
  SamplingMode Sampling = SamplingMode::MAX_INTENSITY;
  double DecayingThreshold = -1;
  bool Ordered = true;
  // calls f(outer index, voxel) for every voxel, in the order above
  template <typename F> void ForEachVoxel(F f) const;

  int To1d(const int v[3]) const { return v[2] * GridSize * GridSize + v[1] * GridSize + v[0]; }
  void To3d(int id, int v[3]) const
  {
    int z = id / (GridSize * GridSize);
    id -= z * GridSize * GridSize;
    int y = id / GridSize;
    id -= y * GridSize;
    v[0] = id; v[1] = y; v[2] = z;
  }
};

template <typename F> void RollingGrid::ForEachVoxel(F f) const
{
  if (!Ordered)
  {
    for (const auto& kvOut : Voxels)
      for (const auto& kvIn : kvOut.second) f(kvOut.first, kvIn.second);
    return;
  }
  struct Ref { unsigned out, in; const Voxel* v; };
  std::vector<Ref> refs;
  refs.reserve(NbPoints);
  for (const auto& kvOut : Voxels)
    for (const auto& kvIn : kvOut.second) refs.push_back({(unsigned)kvOut.first, (unsigned)kvIn.first, &kvIn.second});
  std::sort(refs.begin(), refs.end(), [](const Ref& a, const Ref& b) { return a.out != b.out ? a.out < b.out : a.in < b.in; });
  for (const Ref& r : refs) f((int)r.out, *r.v);
}

}  // namespace orc
