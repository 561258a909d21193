// ORACLE -- TEST INFRASTRUCTURE ONLY (see orc_math.hpp header).  PARITY UNPINNED.
//
// Exact k-nearest-neighbour search standing in for the reference's
// nanoflann::KDTreeSingleIndexAdaptor<L2_Simple_Adaptor<float,...>,...,3,int>
//   slam_lib/include/LidarSlam/KDTreePCLAdaptor.h:35-36, 57-105
// nanoflann (v1.3.2 in the reference CI image) is not under /root/reference;
// its published behaviour is restated:
//   * squared L2 distance evaluated in float as ((dx*dx)+dy*dy)+dz*dz with
//     d = query - point (L2_Simple_Adaptor::evalMetric accumulation order)
//   * result sorted by ascending distance, min(k, N) results
//   * leaf size 16, split on the widest dimension
// Deviation (documented): equal-distance ties are ordered by ascending point
// index instead of tree visit order, which makes the result independent of the
// tree shape (the GPU uses a hash grid, not a kd-tree).
#pragma once
#include <vector>
#include <algorithm>
#include "orc_math.hpp"

namespace orc
{

class KDTree
{
public:
  void Reset(const std::vector<Point>* cloud, int leafMaxSize = 16)
  {
    Cloud = cloud;
    Nodes.clear();
    Idx.clear();
    if (!cloud || cloud->empty()) return;
    Idx.resize(cloud->size());
    for (size_t i = 0; i < Idx.size(); ++i) Idx[i] = (int)i;
    Leaf = leafMaxSize;
    Nodes.reserve(2 * cloud->size() / leafMaxSize + 8);
    Build(0, (int)Idx.size());
  }
  const std::vector<Point>* GetInputCloud() const { return Cloud; }
  bool Empty() const { return !Cloud || Cloud->empty(); }

  // KDTreePCLAdaptor::KnnSearch(const double[3], ...): the query is narrowed to
  // float first (KDTreePCLAdaptor.h:96-100).
  int KnnSearch(const double q[3], int k, int* outIdx, float* outD2) const
  {
    float qf[3] = {(float)q[0], (float)q[1], (float)q[2]};
    return KnnSearch(qf, k, outIdx, outD2);
  }
  int KnnSearch(const float q[3], int k, int* outIdx, float* outD2) const
  {
    if (Empty() || k <= 0) return 0;
    Result r{outIdx, outD2, k, 0};
    Search(0, q, r);
    return r.count;
  }

  static inline float Dist2(const float q[3], const Point& p)
  {
    float r = 0.f;
    float d0 = q[0] - p.x; r += d0 * d0;
    float d1 = q[1] - p.y; r += d1 * d1;
    float d2 = q[2] - p.z; r += d2 * d2;
    return r;
  }

private:
  struct Node { int left, right; int begin, end; int dim; float lo, hi; float bbmin[3], bbmax[3]; };
  struct Result
  {
    int* idx; float* d2; int k; int count;
    float Worst() const { return count < k ? std::numeric_limits<float>::infinity() : d2[k - 1]; }
    void Add(float d, int i)
    {
      // ordered insert by (d, i)
      int pos = count < k ? count : k - 1;
      if (count == k && !(d < d2[k - 1] || (d == d2[k - 1] && i < idx[k - 1]))) return;
      while (pos > 0 && (d < d2[pos - 1] || (d == d2[pos - 1] && i < idx[pos - 1])))
      {
        d2[pos] = d2[pos - 1]; idx[pos] = idx[pos - 1];
        --pos;
      }
      d2[pos] = d; idx[pos] = i;
      if (count < k) ++count;
    }
  };

  int Build(int begin, int end)
  {
    int id = (int)Nodes.size();
    Nodes.emplace_back();
    Node n;
    n.begin = begin; n.end = end; n.left = n.right = -1; n.dim = 0; n.lo = n.hi = 0;
    for (int d = 0; d < 3; ++d) { n.bbmin[d] = std::numeric_limits<float>::max(); n.bbmax[d] = std::numeric_limits<float>::lowest(); }
    for (int i = begin; i < end; ++i)
    {
      const Point& p = (*Cloud)[Idx[i]];
      const float c[3] = {p.x, p.y, p.z};
      for (int d = 0; d < 3; ++d) { n.bbmin[d] = std::min(n.bbmin[d], c[d]); n.bbmax[d] = std::max(n.bbmax[d], c[d]); }
    }
    if (end - begin > Leaf)
    {
      int dim = 0;
      float best = n.bbmax[0] - n.bbmin[0];
      for (int d = 1; d < 3; ++d) if (n.bbmax[d] - n.bbmin[d] > best) { best = n.bbmax[d] - n.bbmin[d]; dim = d; }
      if (best > 0.f)
      {
        int mid = (begin + end) / 2;
        auto coord = [&](int i) { const Point& p = (*Cloud)[i]; return dim == 0 ? p.x : dim == 1 ? p.y : p.z; };
        std::nth_element(Idx.begin() + begin, Idx.begin() + mid, Idx.begin() + end,
                         [&](int a, int b) { float ca = coord(a), cb = coord(b); return ca < cb || (ca == cb && a < b); });
        n.dim = dim;
        Nodes[id] = n;
        int l = Build(begin, mid);
        int r = Build(mid, end);
        Nodes[id].left = l; Nodes[id].right = r;
        return id;
      }
    }
    Nodes[id] = n;
    return id;
  }

  // lower bound of the distance from q to the node's bounding box, in double,
  // shrunk by a relative margin so that float rounding of the per-point
  // distances can never make the search drop a true neighbour.
  static double BoxBound(const Node& n, const float q[3])
  {
    double s = 0;
    for (int d = 0; d < 3; ++d)
    {
      double e = 0;
      if (q[d] < n.bbmin[d]) e = (double)n.bbmin[d] - q[d];
      else if (q[d] > n.bbmax[d]) e = (double)q[d] - n.bbmax[d];
      s += e * e;
    }
    return s * (1.0 - 1e-6);
  }

  void Search(int id, const float q[3], Result& r) const
  {
    const Node& n = Nodes[id];
    if (n.left < 0)
    {
      for (int i = n.begin; i < n.end; ++i) r.Add(Dist2(q, (*Cloud)[Idx[i]]), Idx[i]);
      return;
    }
    double bl = BoxBound(Nodes[n.left], q), br = BoxBound(Nodes[n.right], q);
    int first = n.left, second = n.right;
    double bf = bl, bs = br;
    if (br < bl) { std::swap(first, second); std::swap(bf, bs); }
    if (bf <= (double)r.Worst()) Search(first, q, r);
    if (bs <= (double)r.Worst()) Search(second, q, r);
  }

  const std::vector<Point>* Cloud = nullptr;
  std::vector<Node> Nodes;
  std::vector<int> Idx;
  int Leaf = 16;
};

}  // namespace orc
