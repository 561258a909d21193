// lsa_knn.h -- device-side pieces shared by the matching kernels (lsa_match.hip: staged kNN + model kernels, the
// overlap estimator; lsa_match_fused.hip: one launch per ICP iteration): the view of a target's search grid, the
// 64-bit (distance, index) candidate key and its group minimum, and the per-keypoint model fit.
#pragma once
#include <cfloat>
#include <type_traits>
#include "lsa_ctx.h"
#include "lsa_device_math.h"

namespace lsa
{

__device__ __forceinline__ int cell_coord(float v, float o, float inv, int n)
{
  int c = (int)floorf((v - o) * inv);
  return min(max(c, 0), n - 1);
}

struct GridView
{
  GridDesc g;
  const uint32_t* cell_start;
  const float4* sorted;
  int cx, cy, cz;
  float outd2;
};
__device__ __forceinline__ void grid_view(GridView& v, const GridDesc* desc, const uint32_t* cs, const float4* sorted, float qx, float qy, float qz)
{
  v.g = *desc;
  v.cell_start = cs;
  v.sorted = sorted;
  v.cx = cell_coord(qx, v.g.origin[0], v.g.inv_cell, v.g.dims[0]);
  v.cy = cell_coord(qy, v.g.origin[1], v.g.inv_cell, v.g.dims[1]);
  v.cz = cell_coord(qz, v.g.origin[2], v.g.inv_cell, v.g.dims[2]);
  // squared distance from the query to the grid box (0 inside), shrunk by a guard factor
  const float q[3] = {qx, qy, qz};
  float o = 0.f;
  for (int d = 0; d < 3; ++d)
  {
    const float lo = v.g.origin[d], hi = v.g.origin[d] + v.g.dims[d] * v.g.cell;
    float e = 0.f;
    if (q[d] < lo) e = lo - q[d];
    else if (q[d] > hi) e = q[d] - hi;
    o += e * e;
  }
  v.outd2 = o * 0.999f;
}

struct GridPtrs
{
  const uint32_t* cell_start[kGridLevels];
  const float4* sorted[kGridLevels];
};

// neighbour lists are written SoA: idx[s * cap + q].  knn_cnt[q] = number of neighbours found, or
// kKnnFar when the search proved that the k-th neighbour lies beyond far_d2 (plane / blob matches only
// need to know that: KeypointsMatcher.cxx:217, 303 reject them as NEIGHBORS_TOO_FAR whatever they are).
constexpr int kKnnFar = -1;
// Selection-based search (k_knn_first / k_knn_second).  G lanes cooperate on one query.  The rows of the
// block of cells being searched are contiguous runs of the cell-sorted array; their bounds are fetched by
// as many lanes at once, flattened with a group prefix sum, and the candidates are dealt to the lanes evenly,
// U per lane and batch, all loads in flight together.  The k best of (previous best + batch) are then PICKED:
// k rounds of "group minimum by (distance, index), owner retires it".  Every lane executes the same
// instructions whatever its candidates are -- no per-lane sorted lists, no divergent insertion, no merge
// tree -- and the result sits in registers that are uniform across the group.

// A candidate is ONE 64-bit key: the squared distance's bits above, the target index below.  Distances are
// sums of squares (never negative, never NaN for finite points), so unsigned order of the key IS the search's
// total order (distance, then index) and every comparison of the selection is a single instruction.
typedef unsigned long long knn_key;
constexpr knn_key kKeyEmpty = ((knn_key)0x7f800000u << 32) | 0x7fffffffu;  // (+inf, INT_MAX)
__device__ __forceinline__ knn_key make_key(float d2, int idx) { return ((knn_key)__float_as_uint(d2) << 32) | (unsigned)idx; }
__device__ __forceinline__ float key_d2(knn_key k) { return __uint_as_float((unsigned)(k >> 32)); }
__device__ __forceinline__ int key_idx(knn_key k) { return (int)(unsigned)(k & 0xffffffffu); }

// minimum of the key over the G lanes of a group; result in every lane
template <int G>
__device__ __forceinline__ void group_min(knn_key& m)
{
  auto dpp = [&](auto ctrl) {
    constexpr int c = decltype(ctrl)::value;
    const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)(m & 0xffffffffu), c, 0xF, 0xF, true);
    const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)(m >> 32), c, 0xF, 0xF, true);
    const knn_key o = ((knn_key)hi << 32) | lo;
    m = o < m ? o : m;
  };
  // inside a row of 16 lanes the exchange is a DPP operand modifier (no LDS round trip):
  // quad_perm [1,0,3,2], quad_perm [2,3,0,1], row_half_mirror, row_mirror
  if (G >= 2) dpp(std::integral_constant<int, 0xB1>());
  if (G >= 4) dpp(std::integral_constant<int, 0x4E>());
  if (G >= 8) dpp(std::integral_constant<int, 0x141>());
  if (G >= 16) dpp(std::integral_constant<int, 0x140>());
  // across rows: gfx950's v_permlane16_swap / v_permlane32_swap exchange whole rows between two registers in one
  // VALU pass.  With the same value in both operands, the two results hold (row 0, row 0, row 2, row 2) and
  // (row 1, row 1, row 3, row 3) -- resp. the lower and the upper half twice -- and their minimum is the exchange
  // of the butterfly, without an LDS round trip.
  if (G >= 32)
  {
    const auto lo = __builtin_amdgcn_permlane16_swap((unsigned)(m & 0xffffffffu), (unsigned)(m & 0xffffffffu), false, false);
    const auto hi = __builtin_amdgcn_permlane16_swap((unsigned)(m >> 32), (unsigned)(m >> 32), false, false);
    const knn_key a = ((knn_key)hi[0] << 32) | lo[0], b = ((knn_key)hi[1] << 32) | lo[1];
    m = b < a ? b : a;
  }
  if (G >= 64)
  {
    const auto lo = __builtin_amdgcn_permlane32_swap((unsigned)(m & 0xffffffffu), (unsigned)(m & 0xffffffffu), false, false);
    const auto hi = __builtin_amdgcn_permlane32_swap((unsigned)(m >> 32), (unsigned)(m >> 32), false, false);
    const knn_key a = ((knn_key)hi[0] << 32) | lo[0], b = ((knn_key)hi[1] << 32) | lo[1];
    m = b < a ? b : a;
  }
}

struct MatchConst
{
  int type;
  int k;                 // neighbours requested
  int min_neighbors;     // EdgeMinNbNeighbors
  int single_edge_per_ring;
  double max_dist2;      // MaxNeighborsDistance^2
  double max_model_err;  // Edge/PlaneMaxModelError
  double planarity;
  float ransac_sq_inlier;  // float(EdgeMaxModelError^2)
  int bad_param;           // BAD_MODEL_PARAMETRIZATION for every keypoint
};

__device__ __forceinline__ void write_record(double* __restrict__ rec, int cap, int i, const double A[9], const Vec3<double>& P,
                                             double bx, double by, double bz, double w)
{
#pragma unroll
  for (int f = 0; f < 9; ++f) rec[(size_t)f * cap + i] = A[f];
  rec[(size_t)9 * cap + i] = P.x; rec[(size_t)10 * cap + i] = P.y; rec[(size_t)11 * cap + i] = P.z;
  rec[(size_t)12 * cap + i] = bx; rec[(size_t)13 * cap + i] = by; rec[(size_t)14 * cap + i] = bz;
  rec[(size_t)15 * cap + i] = w;
}

constexpr int kModelBlock = 128;

// One keypoint's match (KeypointsMatcher::BuildLineMatch / BuildPlaneMatch / BuildBlobMatch,
// slam_lib/src/KeypointsMatcher.cxx:106-346) from its n nearest neighbours, ascending (distance, index):
// neighbourhood filter, PCA in double, validity tests, residual record.  nbr_idx(s) / nbr_d2(s) hand out neighbour s
// (static s); nb / nd: LDS staging of the edge candidates, column `col` of rows `nbs` apart.  Returns the status.
template <int KMAX, int TYPE, typename FIdx, typename FD2>
__device__ __forceinline__ int fit_model(const float4 q4, const MatchConst& c, const int n, FIdx nbr_idx, FD2 nbr_d2, const float4* __restrict__ xyzl,
                                         float4* nb, float* nd, const int nbs, const int col, double* __restrict__ rec, const int cap, const int i)
{
  int st = LSA_MATCH_SUCCESS;
  double w = 0.;
  if (c.bad_param)
    st = LSA_MATCH_BAD_MODEL_PARAMETRIZATION;
  else
  {
    const double bx = (double)q4.x, by = (double)q4.y, bz = (double)q4.z;
    Vec3<double> mean, e0, e1, e2;
    double l0 = 0, l1 = 0, l2 = 0;
    CovAccum<double> acc;
    int nsel = 0;
    float last_d2 = 0.f;

    if (TYPE == LSA_EDGE)
    {
      // stage the candidates (ascending distance) in LDS: the filters index them dynamically
#pragma unroll
      for (int s = 0; s < KMAX; ++s)
        if (s < n)
        {
          nb[s * nbs + col] = xyzl[nbr_idx(s)];
          nd[s * nbs + col] = nbr_d2(s);
        }
      if (c.single_edge_per_ring)
      {
        // GetPerRingLineNeighbors (KeypointsMatcher.cxx:349-405): drop the closest point's own ring and
        // rings more than 4 away, then keep the nearest point of every remaining ring
        if (n > 0)
        {
          const int closest = (int)__float_as_uint(nb[col].w);
          for (int t = 0; t < n; ++t)
          {
            const float4 p = nb[t * nbs + col];
            const int lid = (int)__float_as_uint(p.w);
            bool keep = (lid != closest) && (abs(closest - lid) <= 4);
            for (int s = 0; s < t && keep; ++s)
              if ((int)__float_as_uint(nb[s * nbs + col].w) == lid) keep = false;
            if (keep)
            {
              acc.add(p.x, p.y, p.z);
              ++nsel;
              last_d2 = nd[t * nbs + col];
            }
          }
        }
      }
      else
      {
        // GetRansacLineNeighbors (KeypointsMatcher.cxx:408-480)
        if (n >= 2)
        {
          const float4 p1 = nb[col];
          const Vec3<float> P1 = {p1.x, p1.y, p1.z};
          int best = -1, bestCount = 0;
          for (int pi = 1; pi < n; ++pi)
          {
            const float4 p2 = nb[pi * nbs + col];
            const Vec3<float> dir = normalized3(vsub(Vec3<float>{p2.x, p2.y, p2.z}, P1));
            int cnt = 0;
            for (int ci = 1; ci < n; ++ci)
            {
              if (ci == pi) { ++cnt; continue; }
              const float4 pc = nb[ci * nbs + col];
              if (vsqnorm(vcross(vsub(Vec3<float>{pc.x, pc.y, pc.z}, P1), dir)) < c.ransac_sq_inlier) ++cnt;
            }
            if (cnt > bestCount) { bestCount = cnt; best = pi; }
          }
          const float4 pb = nb[best * nbs + col];
          const Vec3<float> dir = normalized3(vsub(Vec3<float>{pb.x, pb.y, pb.z}, P1));
          acc.add(p1.x, p1.y, p1.z);
          nsel = 1;
          last_d2 = nd[col];
          for (int ci = 1; ci < n; ++ci)
          {
            const float4 pc = nb[ci * nbs + col];
            const bool in = (ci == best) || (vsqnorm(vcross(vsub(Vec3<float>{pc.x, pc.y, pc.z}, P1), dir)) < c.ransac_sq_inlier);
            if (in) { acc.add(pc.x, pc.y, pc.z); ++nsel; last_d2 = nd[ci * nbs + col]; }
          }
        }
      }
      if (nsel < c.min_neighbors) st = LSA_MATCH_NOT_ENOUGH_NEIGHBORS;
    }
    else
    {
      // n == kKnnFar: the target holds >= k points but the k-th nearest is beyond MaxNeighborsDistance
      if (n == kKnnFar) st = LSA_MATCH_NEIGHBORS_TOO_FAR;
      else if (n < c.k) st = LSA_MATCH_NOT_ENOUGH_NEIGHBORS;
      else
      {
#pragma unroll
        for (int s = 0; s < KMAX; ++s)
          if (s < c.k)
          {
            const float4 p = xyzl[nbr_idx(s)];
            acc.add(p.x, p.y, p.z);
          }
        nsel = c.k;
#pragma unroll
        for (int s = 0; s < KMAX; ++s)
          if (s == c.k - 1) last_d2 = nbr_d2(s);
      }
    }

    if (st == LSA_MATCH_SUCCESS && (double)last_d2 > c.max_dist2) st = LSA_MATCH_NEIGHBORS_TOO_FAR;
    if (st == LSA_MATCH_SUCCESS)
    {
      Sym3<double> cov;
      acc.finish(nsel, mean, cov);
      eigen33<double>(cov, e0, e1, e2, l0, l1, l2);
      double A[9];
      if (TYPE == LSA_EDGE)
      {
        const double nn[3] = {e2.x, e2.y, e2.z};
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
          for (int b = 0; b < 3; ++b) A[a * 3 + b] = (a == b ? 1. : 0.) - nn[a] * nn[b];
        if (!isfinite(A[0])) st = LSA_MATCH_INVALID_NUMERICAL;
        else
        {
          const double mse = l0 + l1;
          if (mse >= c.max_model_err * c.max_model_err) st = LSA_MATCH_MSE_TOO_LARGE;
          else w = (mse <= 1e-6) ? 1. : 1. - __builtin_sqrt(mse) / c.max_model_err;
        }
      }
      else if (TYPE == LSA_PLANE)
      {
        if (l1 / l2 < c.planarity) st = LSA_MATCH_BAD_PCA_STRUCTURE;
        else
        {
          const double nn[3] = {e0.x, e0.y, e0.z};
#pragma unroll
          for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = 0; b < 3; ++b) A[a * 3 + b] = nn[a] * nn[b];
          if (!isfinite(A[0])) st = LSA_MATCH_INVALID_NUMERICAL;
          else
          {
            const double mse = l0;
            if (mse >= c.max_model_err * c.max_model_err) st = LSA_MATCH_MSE_TOO_LARGE;
            else w = (mse <= 1e-6) ? 1. : 1. - __builtin_sqrt(mse) / c.max_model_err;
          }
        }
      }
      else
      {
        if (l0 <= 0. || l1 <= 0.) st = LSA_MATCH_BAD_PCA_STRUCTURE;
        else
        {
          const double d0 = 1. / __builtin_sqrt(l0), d1 = 1. / __builtin_sqrt(l1), d2 = 1. / __builtin_sqrt(l2);
          const double V[3][3] = {{e0.x, e1.x, e2.x}, {e0.y, e1.y, e2.y}, {e0.z, e1.z, e2.z}};
#pragma unroll
          for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = 0; b < 3; ++b)
              A[a * 3 + b] = ((V[a][0] * d0) * V[b][0] + (V[a][1] * d1) * V[b][1]) + (V[a][2] * d2) * V[b][2];
          if (!isfinite(A[0]) || !isfinite(d0 * d1 * d2)) st = LSA_MATCH_INVALID_NUMERICAL;
          else w = 1.0;
        }
      }
      if (st == LSA_MATCH_SUCCESS) write_record(rec, cap, i, A, mean, bx, by, bz, w);
    }
  }
  if (st != LSA_MATCH_SUCCESS) rec[(size_t)15 * cap + i] = 0.;  // Weights[i] = 0 for rejected keypoints
  return st;
}


}  // namespace lsa
