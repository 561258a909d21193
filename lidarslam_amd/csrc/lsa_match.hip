// lsa_match.hip -- KeypointsMatcher::BuildMatchResiduals and the per-LM-step evaluation of
// the residual blocks on the GPU.
//
//   target grid     replaces KDTreePCLAdaptor::Reset (nanoflann kd-tree build,
//                   slam_lib/include/LidarSlam/KDTreePCLAdaptor.h:57-65) by a three-level dense uniform
//                   grid, all pending targets in one launch sequence: bbox reduce -> cell count (wave-
//                   aggregated atomics on the coarse levels) -> exclusive scan -> cell-sorted float4 copies
//   k_knn_first /   EXACT k-nearest neighbours (KDTreePCLAdaptor::KnnSearch, :79-105).  G lanes per query
//   k_knn_second    search the 3x3x3, then the 5x5x5 block of cells around it; a block's rows are contiguous
//                   runs of the cell-sorted array, flattened by a group prefix sum and dealt evenly to the
//                   lanes; the k best are picked by k rounds of group-minimum (DPP), identical instructions in
//                   every lane.  A round settles the query when k picks lie inside the radius the block
//                   proves; the few percent left go, through a device list, to the second kernel (one
//                   wavefront each, coarser levels, finally the whole target).
//   k_model<..>     one thread per keypoint (slam_lib/src/KeypointsMatcher.cxx:106-346): neighbourhood
//                   filter (per-ring :349-405 / RANSAC line :408-480, candidates staged in LDS), PCA in
//                   double, validity tests, residual record (A, P, X, weight)
//   k_accumulate    what Ceres evaluates per LM step for these blocks
//                   (slam_lib/include/LidarSlam/CeresCostFunctions.h:105-152 + TukeyLoss/ScaledLoss,
//                   KeypointsMatcher.cxx:84-101): cost, g = J^T r, H = J^T J with a fixed-order
//                   wavefront + block + grid reduction (bitwise reproducible run to run)
// kNN order: ascending (float squared distance, target index); the distance is evaluated exactly like
// nanoflann's L2_Simple_Adaptor: ((dx*dx)+dy*dy)+dz*dz with d = query - point.
#include <cfloat>
#include <type_traits>
#include <chrono>
#include <cmath>
#include "lsa_ctx.h"
#include "lsa_device_math.h"
#include "lsa_accum.h"
#include "lsa_knn.h"
#include "lsa_match_internal.h"

using namespace lsa;

namespace lsa
{
InterpConst make_interp_const(const double H0[16], const double H1[16], double t0, double t1);  // lsa_transform.hip
}

namespace
{

__device__ __forceinline__ int f2o(float f)
{
  int i = __float_as_int(f);
  return i >= 0 ? i : i ^ 0x7fffffff;
}
__device__ __forceinline__ float o2f(int i) { return __int_as_float(i >= 0 ? i : i ^ 0x7fffffff); }

// ------------------------------------------------------------------------------------------
// Search-grid construction.  Every target that changed since the last match (the two previous-scan targets
// of the ego-motion step, the two or three sub-maps after a keyframe) is built by ONE sequence of eight
// launches: blockIdx.y selects the target (point passes) or the (target, level) pair (cell passes).
constexpr int kBatchTargets = 6;
struct GridBatch
{
  int ntargets;
  int m[kBatchTargets];
  float cell_hint[kBatchTargets];
  const float4* pts[kBatchTargets];  // AoS points, two float4 per point
  float4* xyzl[kBatchTargets];
  int* bbox[kBatchTargets];
  GridDesc* desc[kBatchTargets];     // [kGridLevels] each
  uint32_t* cell_of[kBatchTargets][kGridLevels];
  uint32_t* cell_start[kBatchTargets][kGridLevels];
  uint32_t* cell_fill[kBatchTargets][kGridLevels];
  uint32_t* block_sums[kBatchTargets][kGridLevels];
  float4* sorted[kBatchTargets][kGridLevels];
};

__global__ __launch_bounds__(256) void k_target_prep(GridBatch gb)
{
  const int t = blockIdx.y;
  const int m = gb.m[t];
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (blockIdx.x * blockDim.x >= m) return;
  float mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
  if (i < m)
  {
    const float4 a = gb.pts[t][2 * (size_t)i];
    const float4 b = gb.pts[t][2 * (size_t)i + 1];
    gb.xyzl[t][i] = make_float4(a.x, a.y, a.z, __uint_as_float(__float_as_uint(b.w) & 0xffffu));
    mn[0] = mx[0] = a.x; mn[1] = mx[1] = a.y; mn[2] = mx[2] = a.z;
  }
  for (int d = 0; d < 3; ++d)
  {
    for (int o = 32; o > 0; o >>= 1)
    {
      mn[d] = fminf(mn[d], __shfl_down(mn[d], o));
      mx[d] = fmaxf(mx[d], __shfl_down(mx[d], o));
    }
  }
  __shared__ float smn[4][3], smx[4][3];
  if ((threadIdx.x & 63) == 0)
    for (int d = 0; d < 3; ++d) { smn[threadIdx.x >> 6][d] = mn[d]; smx[threadIdx.x >> 6][d] = mx[d]; }
  __syncthreads();
  if (threadIdx.x < 3)
  {
    const int d = threadIdx.x;
    atomicMin(&gb.bbox[t][d], f2o(fminf(fminf(smn[0][d], smn[1][d]), fminf(smn[2][d], smn[3][d]))));
    atomicMax(&gb.bbox[t][3 + d], f2o(fmaxf(fmaxf(smx[0][d], smx[1][d]), fmaxf(smx[2][d], smx[3][d]))));
  }
}

// desc[0]: cell = hint (grown until the grid fits its cell budget); every further level has cells 4 x
// larger (grown likewise).  One thread per target.  (Folded into k_grid_zero -- every workgroup deriving the geometry for
// itself -- it saved a launch and cost the zeroing 14 us: the launch covers the cell BUDGET, thousands of workgroups
// that mostly have nothing to zero then all walk the growth loop.)
__global__ void k_grid_setup(GridBatch gb)
{
  const int t = blockIdx.x;
  if (threadIdx.x != 0) return;
  const int* bbox = gb.bbox[t];
  float mn[3], mx[3];
  for (int d = 0; d < 3; ++d) { mn[d] = o2f(bbox[d]); mx[d] = o2f(bbox[3 + d]); }
  float cell = gb.cell_hint[t];
  for (int level = 0; level < kGridLevels; ++level)
  {
    const double cap = (double)grid_level_cells(level);
    if (level > 0) cell *= 4.0f;
    GridDesc g;
    while (true)
    {
      double total = 1;
      for (int d = 0; d < 3; ++d)
      {
        g.dims[d] = (int)floorf((mx[d] - mn[d]) / cell) + 1;
        total *= g.dims[d];
      }
      if (total <= cap) break;
      cell *= 1.26f;
    }
    for (int d = 0; d < 3; ++d) g.origin[d] = mn[d];
    g.cell = cell;
    g.inv_cell = 1.0f / cell;
    g.ncells = g.dims[0] * g.dims[1] * g.dims[2];
    g.npoints = gb.m[t];
    gb.desc[t][level] = g;
  }
}

__global__ __launch_bounds__(256) void k_grid_zero(GridBatch gb)
{
  const int t = blockIdx.y / kGridLevels, l = blockIdx.y % kGridLevels;
  const int nc = gb.desc[t][l].ncells;
  const int i0 = blockIdx.x * 1024 + threadIdx.x;
  if (blockIdx.x * 1024 > nc) return;
  uint32_t* cs = gb.cell_start[t][l];
  uint32_t* cf = gb.cell_fill[t][l];
#pragma unroll
  for (int q = 0; q < 4; ++q)
  {
    const int i = i0 + q * 256;
    if (i <= nc) cs[i] = 0;
    if (i < nc) cf[i] = 0;
  }
}

// Lanes of a wavefront that fall into the same cell are served by ONE atomic: the lowest of them adds the
// group's size, every member learns its rank inside the group.  Keypoints arrive in scan order, so
// neighbouring lanes share the cells of the coarse levels, whose few counters plain atomics would hammer
// from every wave (measured: 72 us per pass on a 33k-point scan, against 10 us).
struct CellGroup
{
  int leader;  // lane that issues the atomic for this lane's cell
  int rank;    // position of this lane among the lanes of its cell
  int count;   // lanes of the wavefront in this cell
};
__device__ __forceinline__ CellGroup group_by_cell(bool active, uint32_t cid)
{
  CellGroup g{-1, 0, 0};
  const int lane = threadIdx.x & 63;
  const unsigned long long below = (1ull << lane) - 1ull;
  unsigned long long remaining = __ballot(active);
  while (remaining)
  {
    const int first = __ffsll((long long)remaining) - 1;
    const uint32_t c = __shfl(cid, first);
    const unsigned long long same = __ballot(active && cid == c);
    if (active && cid == c)
    {
      g.leader = first;
      g.rank = __popcll(same & below);
      g.count = __popcll(same);
    }
    remaining &= ~same;
  }
  return g;
}

// one read of the point, its cell at every level
__global__ __launch_bounds__(256) void k_grid_count(GridBatch gb)
{
  const int t = blockIdx.y;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (blockIdx.x * blockDim.x >= gb.m[t]) return;
  const bool active = i < gb.m[t];
  const float4 p = active ? gb.xyzl[t][i] : make_float4(0.f, 0.f, 0.f, 0.f);
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int l = 0; l < kGridLevels; ++l)
  {
    const GridDesc g = gb.desc[t][l];
    const int cx = cell_coord(p.x, g.origin[0], g.inv_cell, g.dims[0]);
    const int cy = cell_coord(p.y, g.origin[1], g.inv_cell, g.dims[1]);
    const int cz = cell_coord(p.z, g.origin[2], g.inv_cell, g.dims[2]);
    const uint32_t cid = (uint32_t)((cz * g.dims[1] + cy) * g.dims[0] + cx);
    if (active) gb.cell_of[t][l][i] = cid;
    if (l == 0)
    {
      if (active) atomicAdd(&gb.cell_start[t][l][cid], 1u);
    }
    else
    {
      const CellGroup cg = group_by_cell(active, cid);
      if (active && lane == cg.leader) atomicAdd(&gb.cell_start[t][l][cid], (uint32_t)cg.count);
    }
  }
}

// exclusive scan of cell_start[0 .. ncells] in three passes (1024 elements per block)
__global__ __launch_bounds__(256) void k_scan_block(GridBatch gb)
{
  __shared__ uint32_t s[256];
  const int t = blockIdx.y / kGridLevels, l = blockIdx.y % kGridLevels;
  const int total = gb.desc[t][l].ncells + 1;
  const int base = blockIdx.x * 1024;
  if (base >= total) return;
  uint32_t* data = gb.cell_start[t][l];
  uint32_t v[4], tsum = 0;
  for (int q = 0; q < 4; ++q)
  {
    const int i = base + threadIdx.x * 4 + q;
    v[q] = (i < total) ? data[i] : 0;
    tsum += v[q];
  }
  s[threadIdx.x] = tsum;
  __syncthreads();
  for (int o = 1; o < 256; o <<= 1)
  {
    uint32_t a = (threadIdx.x >= (unsigned)o) ? s[threadIdx.x - o] : 0;
    __syncthreads();
    s[threadIdx.x] += a;
    __syncthreads();
  }
  uint32_t run = s[threadIdx.x] - tsum;
  for (int q = 0; q < 4; ++q)
  {
    const int i = base + threadIdx.x * 4 + q;
    if (i < total) data[i] = run;
    run += v[q];
  }
  if (threadIdx.x == 255) gb.block_sums[t][l][blockIdx.x] = s[255];
}
// third pass fused into the second: every block sums the totals of the blocks in front of it itself (a grid of 4 M
// cells is 4 096 blocks: sixteen loads per thread)
__global__ __launch_bounds__(256) void k_scan_add(GridBatch gb)
{
  __shared__ uint32_t part[4];
  const int t = blockIdx.y / kGridLevels, l = blockIdx.y % kGridLevels;
  const int total = gb.desc[t][l].ncells + 1;
  const int base = blockIdx.x * 1024;
  if (base >= total) return;
  const uint32_t* sums = gb.block_sums[t][l];
  uint32_t mine = 0;
  for (int j = threadIdx.x; j < (int)blockIdx.x; j += 256) mine += sums[j];
  for (int o = 32; o > 0; o >>= 1) mine += __shfl_down(mine, o);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = mine;
  __syncthreads();
  const uint32_t add = part[0] + part[1] + part[2] + part[3];
  uint32_t* data = gb.cell_start[t][l];
  for (int q = 0; q < 4; ++q)
  {
    const int i = base + threadIdx.x * 4 + q;
    if (i < total) data[i] += add;
  }
}

__global__ __launch_bounds__(256) void k_grid_scatter(GridBatch gb)
{
  const int t = blockIdx.y;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (blockIdx.x == 0 && threadIdx.x == 0)
  {
    // the bounding box is re-armed for the next build (its last readers were k_grid_zero's workgroups)
    int* bbox = gb.bbox[t];
    for (int d = 0; d < 3; ++d) { bbox[d] = 0x7fffffff; bbox[3 + d] = (int)0x80000000; }
  }
  if (blockIdx.x * blockDim.x >= gb.m[t]) return;
  const bool active = i < gb.m[t];
  const int lane = threadIdx.x & 63;
  const float4 p = active ? gb.xyzl[t][i] : make_float4(0.f, 0.f, 0.f, 0.f);
  const float4 rec = make_float4(p.x, p.y, p.z, __int_as_float(i));
#pragma unroll
  for (int l = 0; l < kGridLevels; ++l)
  {
    const uint32_t cid = active ? gb.cell_of[t][l][i] : 0u;
    uint32_t slot;
    if (l == 0)
      slot = active ? atomicAdd(&gb.cell_fill[t][l][cid], 1u) : 0u;
    else
    {
      const CellGroup cg = group_by_cell(active, cid);
      uint32_t base = 0;
      if (active && lane == cg.leader) base = atomicAdd(&gb.cell_fill[t][l][cid], (uint32_t)cg.count);
      slot = __shfl(base, max(cg.leader, 0)) + (uint32_t)cg.rank;
    }
    if (active) gb.sorted[t][l][gb.cell_start[t][l][cid] + slot] = rec;
  }
}

// ------------------------------------------------------------------------------------------

template <int KMAX, int G, int U>
struct GroupSelect
{
  static constexpr int C = (KMAX + G - 1) / G;  // carry slots per lane: the previous best, dealt over the group
  knn_key key[U + C];   // [0, U) fresh candidates of the batch, [U, U + C) carry
  knn_key best[KMAX];   // ascending, uniform across the group; kKeyEmpty = none
  __device__ __forceinline__ void reset()
  {
#pragma unroll
    for (int s = 0; s < KMAX; ++s) best[s] = kKeyEmpty;
#pragma unroll
    for (int u = 0; u < U + C; ++u) key[u] = kKeyEmpty;
  }
  // best <- the k smallest of (carry slots + the fresh candidates of every lane); nfresh: fresh slots any group
  // of the wavefront uses in this batch (wave-uniform; the others hold nothing and are not looked at)
  __device__ __forceinline__ void select(int k, int gl, int nfresh)
  {
    // nothing in this batch beats the current k-th best of any group of the wavefront: keep the list
    knn_key kth = kKeyEmpty;
#pragma unroll
    for (int s = 0; s < KMAX; ++s)
      if (s == k - 1) kth = best[s];
    bool improves = false;
#pragma unroll
    for (int u = 0; u < U; ++u) improves |= key[u] < kth;
    if (!__any(improves)) return;
#pragma unroll
    for (int s = 0; s < KMAX; ++s)
    {
      if (s < k)
      {
        knn_key m = key[U];
#pragma unroll
        for (int c = 1; c < C; ++c) m = key[U + c] < m ? key[U + c] : m;
#pragma unroll
        for (int u = 0; u < U; ++u)
          if (u < nfresh) m = key[u] < m ? key[u] : m;
        group_min<G>(m);
        best[s] = m;
        // the owner retires it (keys of real candidates are unique; empty slots all look alike, harmless)
#pragma unroll
        for (int c = 0; c < C; ++c)
          if (key[U + c] == m) key[U + c] = kKeyEmpty;
#pragma unroll
        for (int u = 0; u < U; ++u)
          if (u < nfresh && key[u] == m) key[u] = kKeyEmpty;
      }
    }
    // the new best becomes the carry of the next batch: entry s lives in slot s / G of lane s % G
#pragma unroll
    for (int c = 0; c < C; ++c) key[U + c] = kKeyEmpty;
#pragma unroll
    for (int s = 0; s < KMAX; ++s)
      if (s < k && gl == s % G) key[U + s / G] = best[s];
  }
  __device__ __forceinline__ int count_below(float bound2, int k) const
  {
    int c = 0;
#pragma unroll
    for (int s = 0; s < KMAX; ++s)
      if (s < k && key_d2(best[s]) < bound2) ++c;
    return c;
  }
};

// the rows of one block of cells, dealt E per lane and flattened: candidate c of [0, total) is an offset into
// the cell-sorted array
template <int G, int E>
struct BlockRuns
{
  static constexpr int kE = E;
  uint32_t b[E], len[E];
  uint32_t excl, total;
  bool covered;
  // every lane of the wavefront calls these (shuffles); groups with live == false get an empty block.
  // fetch() only issues the loads of the row bounds, finish() consumes them: several blocks can be fetched
  // before the first is finished, their loads overlap.
  __device__ __forceinline__ void fetch(const GridView& gv, int r, int gl, bool live)
  {
    const int nx = gv.g.dims[0], ny = gv.g.dims[1], nz = gv.g.dims[2];
    const int z0 = max(0, gv.cz - r), z1 = min(nz - 1, gv.cz + r);
    const int y0 = max(0, gv.cy - r), y1 = min(ny - 1, gv.cy + r);
    const int x0 = max(0, gv.cx - r), x1 = min(nx - 1, gv.cx + r);
    const int ys = y1 - y0 + 1;
    const int nrows = live ? (z1 - z0 + 1) * ys : 0;
    covered = (x0 == 0 && y0 == 0 && z0 == 0 && x1 == nx - 1 && y1 == ny - 1 && z1 == nz - 1);
    const int inv_ys = (1 << 16) / ys + 1;  // ri / ys == (ri * inv_ys) >> 16 for ys <= 9, ri < 128: no integer division in the loop
#pragma unroll
    for (int e = 0; e < E; ++e)
    {
      const int ri = gl * E + e;
      b[e] = 0; len[e] = 0;
      if (ri < nrows)
      {
        const int zi = (ri * inv_ys) >> 16;
        const int row = ((z0 + zi) * ny + (y0 + ri - zi * ys)) * nx;
        b[e] = gv.cell_start[row + x0];
        len[e] = gv.cell_start[row + x1 + 1];  // end of the run until finish()
      }
    }
  }
  __device__ __forceinline__ void finish(int gl)
  {
    uint32_t mine = 0;
#pragma unroll
    for (int e = 0; e < E; ++e)
    {
      len[e] -= b[e];
      mine += len[e];
    }
    uint32_t inc = mine;
#pragma unroll
    for (int o = 1; o < G; o <<= 1)
    {
      const uint32_t t = __shfl_up(inc, o, G);
      if (gl >= o) inc += t;
    }
    excl = inc - mine;  // non-decreasing over the lanes of the group
    total = __shfl(inc, G - 1, G);
  }
  __device__ __forceinline__ void build(const GridView& gv, int r, int gl, bool live)
  {
    fetch(gv, r, gl, live);
    finish(gl);
  }
  // the whole target as one run (exhaustive stage)
  __device__ __forceinline__ void whole(uint32_t m, int gl)
  {
#pragma unroll
    for (int e = 0; e < E; ++e) { b[e] = 0; len[e] = 0; }
    if (gl == 0) len[0] = m;
    excl = gl == 0 ? 0u : m;
    total = m;
    covered = true;
  }
  __device__ __forceinline__ uint32_t locate(uint32_t c) const
  {
    // the last lane L of the group with excl[L] <= c owns candidate c (then c < excl[L + 1]: it has a non-empty row)
    int L = 0;
#pragma unroll
    for (int step = G / 2; step > 0; step >>= 1)
    {
      const uint32_t ex = __shfl(excl, L + step, G);
      if (ex <= c) L += step;
    }
    uint32_t off = c - __shfl(excl, L, G);
    if constexpr (E == 1) return __shfl(b[0], L, G) + off;  // the lane's only row holds it
    uint32_t addr = 0;
    bool found = false;
#pragma unroll
    for (int e = 0; e < E; ++e)
    {
      const uint32_t bb = __shfl(b[e], L, G), ll = __shfl(len[e], L, G);
      if (!found && off < ll) { addr = bb + off; found = true; }
      if (!found) off -= ll;
    }
    return addr;
  }
};

// one search round: the k best of the block's candidates end up in sel.best (uniform across the group)
template <int KMAX, int G, int U, int E>
__device__ __forceinline__ void search_block(GroupSelect<KMAX, G, U>& sel, const BlockRuns<G, E>& runs, const float4* __restrict__ sorted, int k, int gl,
                                             float qx, float qy, float qz)
{
  sel.reset();
  // software pipeline: the loads of batch i + 1 are in flight while batch i is being picked from
  float4 nxt[U];
  auto issue = [&](uint32_t base) {
#pragma unroll
    for (int u = 0; u < U; ++u)
    {
      // a slot no group of the wavefront has a candidate for costs nothing (most blocks fill one or two slots)
      if (!__any(base + u * G + gl < runs.total)) break;
      const uint32_t c = base + u * G + gl;
      const uint32_t addr = runs.locate(c);
      nxt[u] = sorted[c < runs.total ? addr : 0];
    }
  };
  if (__any(0u < runs.total)) issue(0);
  for (uint32_t base = 0; __any(base < runs.total); base += G * U)
  {
    float4 p[U];
#pragma unroll
    for (int u = 0; u < U; ++u) p[u] = nxt[u];
    if (__any(base + G * U < runs.total)) issue(base + G * U);
    int nfresh = 0;
#pragma unroll
    for (int u = 0; u < U; ++u)
    {
      const bool ok = base + u * G + gl < runs.total;
      if (__any(ok)) nfresh = u + 1;
      const float dx = qx - p[u].x, dy = qy - p[u].y, dz = qz - p[u].z;
      sel.key[u] = ok ? make_key((dx * dx + dy * dy) + dz * dz, __float_as_int(p[u].w)) : kKeyEmpty;
    }
    sel.select(k, gl, nfresh);
  }
}

// First stage: every query, G lanes each, the 3x3x3, then the 5x5x5 (and with RMAX = 3 the 7x7x7) block of the
// finest grid (each round searches its whole block afresh: no bookkeeping of what the previous round saw).
// Queries that are not settled inside RMAX cells go to the second stage through the device list.
template <int KMAX, int G, int RMAX>
__global__ __launch_bounds__(256) void k_knn_first(const float4* __restrict__ queries, int nq, Rigid pose, int k, float far_d2,
                                                   const GridDesc* __restrict__ desc, GridPtrs gp, int* __restrict__ knn_idx,
                                                   float* __restrict__ knn_d2, int* __restrict__ knn_cnt, int cap, int* __restrict__ count_out,
                                                   int* __restrict__ list_out, float4* __restrict__ list_pts)
{
  constexpr int U = 4;
  const int gl = threadIdx.x % G;
  const int q = (int)(((size_t)blockIdx.x * 256 + threadIdx.x) / G);
  const bool active = q < nq;
  float qx = 0.f, qy = 0.f, qz = 0.f;
  if (active)
  {
    // KeypointsMatcher: worldPoint = PosePrior * basePoint in double, narrowed to float for the search
    const float4 q4 = queries[2 * (size_t)q];
    double wx, wy, wz;
    rigid_apply(pose, (double)q4.x, (double)q4.y, (double)q4.z, wx, wy, wz);
    qx = (float)wx; qy = (float)wy; qz = (float)wz;
  }
  GridView gv;
  grid_view(gv, desc, gp.cell_start[0], gp.sorted[0], qx, qy, qz);
  GroupSelect<KMAX, G, U> sel;
  sel.reset();
  bool done = !active, far = false, deferred = false;
  // the row bounds of every round's block in one memory round trip (a later round costs one trip less); each
  // block has its own number of rows per lane, so that locating a candidate in the 3x3x3 block costs no more
  // shuffles than its 9 rows need
  constexpr int E1 = (9 + G - 1) / G, E2 = (25 + G - 1) / G, E3 = (49 + G - 1) / G;
  BlockRuns<G, E1> runs1;
  BlockRuns<G, E2> runs2;
  BlockRuns<G, (RMAX >= 3 ? E3 : 1)> runs3;
  runs1.fetch(gv, 1, gl, active);
  runs2.fetch(gv, 2, gl, active);
  if (RMAX >= 3) runs3.fetch(gv, 3, gl, active);
  auto round = [&](auto& runs, int r) {
    if (__all(done)) return;
    runs.finish(gl);
    if (done) runs.total = 0;  // groups that are done keep their result: an empty block, `cur` is scratch for them
    GroupSelect<KMAX, G, U> cur;
    search_block<KMAX, G, U, std::remove_reference_t<decltype(runs)>::kE>(cur, runs, gv.sorted, k, gl, qx, qy, qz);
    if (done) return;
    sel = cur;
    // every point closer than r cells (minus a 0.1 % guard for the float cell assignment) has been seen
    const float br = ((float)r - 0.001f) * gv.g.cell;
    const float bound2 = gv.outd2 + br * br;
    if (runs.covered || sel.count_below(bound2, k) >= k) done = true;
    else if (bound2 > far_d2) { far = true; done = true; }
    else if (r == RMAX)
    {
      // handed to the second stage: the query in target coordinates, and an upper bound of the k-th distance
      // (the k-th best seen so far; +inf when the block holds fewer than k points)
      if (gl == 0)
      {
        float ub = INFINITY;
#pragma unroll
        for (int s = 0; s < KMAX; ++s)
          if (s == k - 1) ub = key_d2(sel.best[s]);
        const int slot = atomicAdd(count_out, 1);
        list_out[slot] = q;
        list_pts[slot] = make_float4(qx, qy, qz, ub);
      }
      deferred = true;
      done = true;
    }
  };
  round(runs1, 1);
  round(runs2, 2);
  if (RMAX >= 3) round(runs3, 3);
  if (active && gl == 0 && !deferred)
  {
    int cnt = 0;
#pragma unroll
    for (int s = 0; s < KMAX; ++s)
      if (s < k)
      {
        knn_idx[(size_t)s * cap + q] = key_idx(sel.best[s]);
        knn_d2[(size_t)s * cap + q] = key_d2(sel.best[s]);
        if (sel.best[s] != kKeyEmpty) ++cnt;
      }
    knn_cnt[q] = far ? kKnnFar : cnt;
  }
}

// Second and last stage: one wavefront per query the first stage handed over (a few percent: the isolated
// keypoints).  Coarser levels, blocks of 3^3, 5^3, 7^3 cells each, finally the whole target as one
// run, so every query leaves this kernel answered.  Same (distance, index) order everywhere => the result
// does not depend on the route taken.
template <int KMAX>
__global__ __launch_bounds__(256) void k_knn_second(const int* __restrict__ list_in, const float4* __restrict__ list_pts, const int* __restrict__ count_in,
                                                    int list_cap, int k, float far_d2, const GridDesc* __restrict__ desc, GridPtrs gp,
                                                    int* __restrict__ knn_idx, float* __restrict__ knn_d2, int* __restrict__ knn_cnt, int cap,
                                                    int* __restrict__ exhaustive_count)
{
  constexpr int G = 64, U = 8, E = 1;
  constexpr int kStages = 3 * (kGridLevels - 1);  // blocks (level 1, r = 1 .. 3), (level 2, r = 1 .. 3); then the whole target
  const int gl = threadIdx.x & 63;
  const int nwaves = gridDim.x * 4;
  // the list entry is loaded together with the count (its slot exists whatever the count is): one round trip
  int w = blockIdx.x * 4 + (threadIdx.x >> 6);
  int q = list_in[min(w, list_cap - 1)];
  float4 qp = list_pts[min(w, list_cap - 1)];
  const int nwork = *count_in;
  for (; w < nwork; w += nwaves, q = list_in[min(w, list_cap - 1)], qp = list_pts[min(w, list_cap - 1)])
  {
    const float qx = qp.x, qy = qp.y, qz = qp.z;
    GroupSelect<KMAX, G, U> sel;
    sel.reset();
    bool done = false, far = false;
    GridView gv1, gv2;
    grid_view(gv1, desc + 1, gp.cell_start[1], gp.sorted[1], qx, qy, qz);
    grid_view(gv2, desc + 2, gp.cell_start[2], gp.sorted[2], qx, qy, qz);
    // every point closer than r cells (minus a 0.1 % guard for the float cell assignment) is in block (level, r)
    auto proven = [&](int stage) {
      const GridView& gv = stage < 3 ? gv1 : gv2;
      const float br = ((float)(1 + stage % 3) - 0.001f) * gv.g.cell;
      return gv.outd2 + br * br;
    };
    // Upper bound of the k-th distance (+inf: none yet), from the first stage and then from every scan that did
    // not settle the query: the first block whose proven radius exceeds it settles the query for certain, so the
    // search starts there -- one fetch of row bounds, one scan.  Without a bound the blocks are tried in order;
    // one that holds fewer than k points is not scanned.  The last "block" is the whole target.
    float ub = qp.w;
    int stage = 0;
    if (ub != INFINITY)
      while (stage < kStages - 1 && !(proven(stage) > ub)) ++stage;
#pragma unroll 1
    for (; stage <= kStages && !done; ++stage)
    {
      BlockRuns<G, E> runs;
      float bound2 = INFINITY;
      const float4* src = gp.sorted[0];
      if (stage < kStages)
      {
        const bool l1 = stage < 3;
        GridView gv;
        gv.g = l1 ? gv1.g : gv2.g;
        gv.cell_start = l1 ? gv1.cell_start : gv2.cell_start;
        gv.sorted = l1 ? gv1.sorted : gv2.sorted;
        gv.cx = l1 ? gv1.cx : gv2.cx; gv.cy = l1 ? gv1.cy : gv2.cy; gv.cz = l1 ? gv1.cz : gv2.cz;
        gv.outd2 = l1 ? gv1.outd2 : gv2.outd2;
        runs.build(gv, 1 + stage % 3, gl, true);
        bound2 = proven(stage);
        src = gv.sorted;
      }
      else
      {
        runs.whole((uint32_t)desc->npoints, gl);
        if (gl == 0) atomicAdd(exhaustive_count, 1);
      }
      const bool few = runs.total < (uint32_t)k;  // cannot hold k neighbours
      if (runs.covered || (!few && (ub == INFINITY || bound2 > ub || stage == kStages - 1)))
      {
        search_block<KMAX, G, U, E>(sel, runs, src, k, gl, qx, qy, qz);
        if (runs.covered || sel.count_below(bound2, k) >= k) done = true;
        else if (bound2 > far_d2) { far = true; done = true; }
        else
        {
#pragma unroll
          for (int s = 0; s < KMAX; ++s)
            if (s == k - 1) ub = key_d2(sel.best[s]);
        }
      }
      // fewer than k points inside a radius beyond the rejection distance
      else if (few && bound2 > far_d2) { far = true; done = true; }
    }
    if (gl == 0)
    {
      int cnt = 0;
#pragma unroll
      for (int s = 0; s < KMAX; ++s)
        if (s < k)
        {
          knn_idx[(size_t)s * cap + q] = key_idx(sel.best[s]);
          knn_d2[(size_t)s * cap + q] = key_d2(sel.best[s]);
          if (sel.best[s] != kKeyEmpty) ++cnt;
        }
      knn_cnt[q] = far ? kKnnFar : cnt;
    }
  }
}

// ------------------------------------------------------------------------------------------
template <int KMAX, int TYPE>
__global__ __launch_bounds__(kModelBlock) void k_model(const float4* __restrict__ queries, int nq, MatchConst c, const int* __restrict__ knn_idx,
                                                       const float* __restrict__ knn_d2, const int* __restrict__ knn_cnt,
                                                       const float4* __restrict__ xyzl, double* __restrict__ rec,
                                                       uint8_t* __restrict__ status, int cap, int* __restrict__ hist)
{
  __shared__ int lh[LSA_MATCH_NSTATUS];
  __shared__ float4 nb[(TYPE == LSA_EDGE ? KMAX : 1) * kModelBlock];  // edge candidates staged in LDS
  __shared__ float nd[(TYPE == LSA_EDGE ? KMAX : 1) * kModelBlock];
  if (threadIdx.x < LSA_MATCH_NSTATUS) lh[threadIdx.x] = 0;
  __syncthreads();
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < nq)
  {
    const int n = c.bad_param ? 0 : knn_cnt[i];
    const int st = fit_model<KMAX, TYPE>(
      queries[2 * (size_t)i], c, n, [&](int s) { return knn_idx[(size_t)s * cap + i]; }, [&](int s) { return knn_d2[(size_t)s * cap + i]; }, xyzl, nb, nd,
      kModelBlock, threadIdx.x, rec, cap, i);
    status[i] = (uint8_t)st;
    atomicAdd(&lh[st], 1);
  }
  __syncthreads();
  if (threadIdx.x < LSA_MATCH_NSTATUS && lh[threadIdx.x]) atomicAdd(&hist[threadIdx.x], lh[threadIdx.x]);
}

__global__ void k_fill_status(uint8_t* __restrict__ status, double* __restrict__ rec, int cap, int n, uint8_t v)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) { status[i] = v; rec[(size_t)15 * cap + i] = 0.; }
}

// ------------------------------------------------------------------------------------------
// One evaluation of the residual blocks (lsa_accum.h) per launch: the host-driven trust region (lsa_accumulate).
__global__ __launch_bounds__(256) void k_accumulate(AccumConst c, double* __restrict__ partials, unsigned long long* __restrict__ mailbox, unsigned tag)
{
  double acc[kAccumVals];
#pragma unroll
  for (int v = 0; v < kAccumVals; ++v) acc[v] = 0.;
  accumulate_records(c.set, c.rot, c.jac != 0, blockIdx.x * blockDim.x + threadIdx.x, gridDim.x * blockDim.x, acc);
  // fixed-order reduction: wavefront (transposed: permlane swaps + DPP, lsa_accum.h), then the 4 waves through LDS
  __shared__ double wsum[4][kAccumVals];
  int slot;
  const double total = wave_reduce_accum(acc, slot);  // this lane's one value of the 29, summed over the wavefront
  if ((threadIdx.x & 1) == 0 && slot < kAccumVals) wsum[threadIdx.x >> 6][slot] = total;
  __syncthreads();
  // Zero-copy hand-over: the block's 29 partial sums land in coherent host memory as 58 granules.  A granule is
  // ONE naturally aligned 8-byte word -- the evaluation's tag above, one half of a double below -- written by ONE
  // relaxed system-scope atomic store: tag and payload are the same memory object, so the payload can never be
  // seen without its tag, whatever order the fabric delivers the lanes' stores in.  The host polls the granules
  // (relaxed 64-bit atomic loads) and folds the blocks in index order (as k_accumulate_final does): no fence, no
  // flag, no second kernel, no D2H copy, no stream synchronisation on the LM critical path.
  if (threadIdx.x < 2 * kAccumVals)
  {
    const int v = threadIdx.x >> 1, half = threadIdx.x & 1;
    const double r = ((wsum[0][v] + wsum[1][v]) + wsum[2][v]) + wsum[3][v];
    if (half == 0) partials[(size_t)blockIdx.x * kAccumVals + v] = r;
    if (mailbox)
    {
      const unsigned long long bits = (unsigned long long)__double_as_longlong(r);
      const unsigned word = half ? (unsigned)(bits >> 32) : (unsigned)(bits & 0xffffffffull);
      __hip_atomic_store(mailbox + (size_t)blockIdx.x * kMailboxStride + threadIdx.x, ((unsigned long long)tag << 32) | word, __ATOMIC_RELAXED,
                         __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

// folds the per-block partials in block order (the path without a mailbox; same order as the host fold)
__global__ void k_accumulate_final(const double* __restrict__ partials, int nblocks, double* __restrict__ out)
{
  const int v = threadIdx.x;
  if (v >= kAccumVals) return;
  double s = 0.;
  for (int b = 0; b < nblocks; ++b) s += partials[(size_t)b * kAccumVals + v];
  out[v] = s;
}

__global__ void k_records_to_aos(const double* __restrict__ rec, const uint8_t* __restrict__ status, int cap, int n, double* __restrict__ out,
                                 double* __restrict__ weights)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const bool ok = status[i] == LSA_MATCH_SUCCESS;
  if (out)
    for (int f = 0; f < 16; ++f) out[(size_t)i * 16 + f] = ok ? rec[(size_t)f * cap + i] : 0.;
  weights[i] = ok ? rec[(size_t)15 * cap + i] : 0.;
}

// ---- Confidence::LCPEstimator (slam_lib/src/ConfidenceEstimators.cxx:27-65, Slam::EstimateOverlap Slam.cxx:1370-1388) ----
__device__ __forceinline__ double point_time(const float4& b) { return __hiloint2double(__float_as_int(b.y), __float_as_int(b.x)); }
// sampled points of the frame, registered into the world (undistorted when asked), as kNN queries
__global__ __launch_bounds__(256) void k_overlap_queries(const float4* __restrict__ frame, int nb, float ratio, int interpolate, InterpConst c, Rigid R,
                                                         float4* __restrict__ out)
{
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= nb) return;
  const size_t src = (size_t)((float)n / ratio);  // cloud->at(n / subsamplingRatio)
  float4 a = frame[2 * src];
  const float4 b = frame[2 * src + 1];
  Rigid T = R;
  if (interpolate) interp_eval(c, point_time(b), T);
  double ox, oy, oz;
  rigid_apply(T, (double)a.x, (double)a.y, (double)a.z, ox, oy, oz);
  a.x = (float)ox; a.y = (float)oy; a.z = (float)oz;
  out[2 * (size_t)n] = a;
  out[2 * (size_t)n + 1] = b;
}
// best Gaussian score over the maps per point, summed: block partials in a fixed order
struct OverlapConst
{
  const float* d2[3];   // nearest squared distance per sampled point (slot 0 of the kNN output), nullptr = map not used
  float inv2sq[3];      // 1 / (2 (leaf / 3)^2)
};
__global__ __launch_bounds__(256) void k_overlap_score(OverlapConst c, int nb, float* __restrict__ partials)
{
  __shared__ float ws[4];
  float acc = 0.f;
  for (int n = blockIdx.x * blockDim.x + threadIdx.x; n < nb; n += gridDim.x * blockDim.x)
  {
    float best = 0.f;
#pragma unroll
    for (int k = 0; k < 3; ++k)
      if (c.d2[k])
      {
        const float p = expf(-c.d2[k][n] * c.inv2sq[k]);
        best = p > best ? p : best;
      }
    acc += best;
  }
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o);
  if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) partials[blockIdx.x] = ((ws[0] + ws[1]) + ws[2]) + ws[3];
}

// builds the search grids of the listed targets, all in one sequence of launches on `st`
static int build_grids(lsa_ctx* ctx, const int* tis, int count, hipStream_t st)
{
  GridBatch gb;
  int nt = 0, max_m = 0, max_cells = 0;
  double bytes = 0;
  for (int i = 0; i < count; ++i)
  {
    Target& t = ctx->target[tis[i]];
    t.dirty = false;
    if (t.m == 0) continue;
    gb.m[nt] = t.m;
    gb.cell_hint[nt] = t.cell_hint;
    gb.pts[nt] = reinterpret_cast<const float4*>(t.pts);
    gb.xyzl[nt] = t.xyzl;
    gb.bbox[nt] = t.bbox_bits;
    gb.desc[nt] = t.desc;
    for (int l = 0; l < kGridLevels; ++l)
    {
      gb.cell_of[nt][l] = t.lv[l].cell_of;
      gb.cell_start[nt][l] = t.lv[l].cell_start;
      gb.cell_fill[nt][l] = t.lv[l].cell_fill;
      gb.block_sums[nt][l] = t.lv[l].block_sums;
      gb.sorted[nt][l] = t.lv[l].sorted;
      max_cells = std::max(max_cells, t.lv[l].max_cells);
    }
    max_m = std::max(max_m, t.m);
    bytes += (double)t.m * (32 + 16 + kGridLevels * (16 + 4 + 4 + 16 + 16));
    ++nt;
  }
  if (nt == 0) return LSA_OK;
  gb.ntargets = nt;
  ProfScope ps(ctx, st == ctx->stream ? "target_grid_build" : "target_grid_build_ahead", bytes, st);
  const int pb = (max_m + 255) / 256;
  const int cb = (max_cells + 1 + 1023) / 1024;  // the cell passes return at once beyond a grid's own cell count
  hipLaunchKernelGGL(k_target_prep, dim3(pb, nt), dim3(256), 0, st, gb);
  hipLaunchKernelGGL(k_grid_setup, dim3(nt), dim3(64), 0, st, gb);
  hipLaunchKernelGGL(k_grid_zero, dim3(cb, nt * kGridLevels), dim3(256), 0, st, gb);
  hipLaunchKernelGGL(k_grid_count, dim3(pb, nt), dim3(256), 0, st, gb);
  hipLaunchKernelGGL(k_scan_block, dim3(cb, nt * kGridLevels), dim3(256), 0, st, gb);
  hipLaunchKernelGGL(k_scan_add, dim3(cb, nt * kGridLevels), dim3(256), 0, st, gb);
  hipLaunchKernelGGL(k_grid_scatter, dim3(pb, nt), dim3(256), 0, st, gb);
  return LSA_OK;
}

// ... of every target marked dirty
int flush_grids(lsa_ctx* ctx)
{
  int tis[6], n = 0;
  for (int ti = 0; ti < 6; ++ti)
    if (ctx->target[ti].dirty) tis[n++] = ti;
  return n ? build_grids(ctx, tis, n, ctx->stream) : LSA_OK;
}

template <int KMAX>
void launch_knn(lsa_ctx* ctx, const lsa_point_t* q, int nq, const Rigid& pose, int k, float far_d2, int type, int ti, hipStream_t st, int* hist)
{
  Target& t = ctx->target[ti];
  MatchBuf& mb = ctx->match[type];
  GridPtrs gp;
  for (int l = 0; l < kGridLevels; ++l) { gp.cell_start[l] = t.lv[l].cell_start; gp.sorted[l] = t.lv[l].sorted; }
  const float4* q4 = reinterpret_cast<const float4*>(q);
  int* cntA = hist + LSA_MATCH_NSTATUS;  // queries handed from the first to the second stage
  int* cntB = cntA + 1;                  // queries that needed the exhaustive scan (diagnostics)
  int* listA = mb.slow_list;
  const char* nf = type == LSA_EDGE ? "knn_fine_edge" : type == LSA_PLANE ? "knn_fine_plane" : "knn_fine_blob";
  const char* nc = type == LSA_EDGE ? "knn_coarse_edge" : type == LSA_PLANE ? "knn_coarse_plane" : "knn_coarse_blob";
  {
    // algorithmic bytes: query point in, k candidate points examined at least, k (index, distance) pairs out
    ProfScope ps(ctx, nf, (double)nq * (32 + k * 16 + k * 8), st);
    const int lanes = ctx->knn_lanes[type];
    const int rounds = ctx->knn_rounds[type];
#define LSA_FIRST(G, R)                                                                                                                     \
  hipLaunchKernelGGL((k_knn_first<KMAX, G, R>), dim3((int)(((size_t)nq * G + 255) / 256)), dim3(256), 0, st, q4, nq, pose, k, far_d2, t.desc, gp, \
                     mb.knn_idx, mb.knn_d2, mb.knn_cnt, mb.cap, cntA, listA, mb.slow_pts)
    if (lanes >= 32) { if (rounds >= 3) LSA_FIRST(32, 3); else LSA_FIRST(32, 2); }
    else if (lanes >= 16) { if (rounds >= 3) LSA_FIRST(16, 3); else LSA_FIRST(16, 2); }
    else { if (rounds >= 3) LSA_FIRST(8, 3); else LSA_FIRST(8, 2); }
#undef LSA_FIRST
  }
  {
    // the deferred share is only known on the device: no bytes are credited to this stage
    ProfScope ps(ctx, nc, 0., st);
    hipLaunchKernelGGL((k_knn_second<KMAX>), dim3(512), dim3(256), 0, st, (const int*)listA, (const float4*)mb.slow_pts, (const int*)cntA, mb.cap, k,
                       far_d2, t.desc, gp, mb.knn_idx, mb.knn_d2, mb.knn_cnt, mb.cap, cntB);
  }
}

template <int KMAX, int TYPE>
void launch_model(lsa_ctx* ctx, const lsa_point_t* q, int nq, const MatchConst& mc, int type, int ti, hipStream_t st, int* hist)
{
  Target& t = ctx->target[ti];
  MatchBuf& mb = ctx->match[type];
  hipLaunchKernelGGL((k_model<KMAX, TYPE>), dim3((nq + kModelBlock - 1) / kModelBlock), dim3(kModelBlock), 0, st,
                     reinterpret_cast<const float4*>(q), nq, mc, mb.knn_idx, mb.knn_d2, mb.knn_cnt, t.xyzl, mb.rec, mb.status, mb.cap, hist);
}

}  // namespace

namespace lsa
{
int build_target_grids(lsa_ctx* ctx, const int* tis, int count, hipStream_t st) { return build_grids(ctx, tis, count, st); }
}  // namespace lsa

extern "C" {

int lsa_set_target(lsa_ctx* ctx, int slot, int type, const lsa_point_t* pts, int m)
{
  if (!ctx || slot < 0 || slot > 1 || type < 0 || type > 2 || m < 0 || (!pts && m > 0)) return ctx ? ctx->fail(LSA_E_ARG, "lsa_set_target: bad argument") : LSA_E_ARG;
  LSA_HIP(ctx, hipSetDevice(ctx->device));
  const int ti = slot * 3 + type;
  int rc = ensure_target(ctx, ti, m);
  if (rc) return rc;
  Target& t = ctx->target[ti];
  t.m = m;
  if (m == 0) return LSA_OK;
  {
    ProfScope ps(ctx, "target_upload_h2d", (double)m * 32);
    LSA_HIP(ctx, hipMemcpyAsync(t.pts, pts, (size_t)m * sizeof(lsa_point_t), hipMemcpyHostToDevice, ctx->stream));
  }
  t.dirty = true;  // the search grid is built with the other pending targets at the next match
  LSA_HIP(ctx, hipStreamSynchronize(ctx->stream));  // the host buffer may be pageable and go away
  return LSA_OK;
}

lsa_point_t* lsa_target_staging(lsa_ctx* ctx, int slot, int type, int capacity)
{
  if (!ctx || slot < 0 || slot > 1 || type < 0 || type > 2 || capacity < 0) return nullptr;
  const int ti = slot * 3 + type;
  if (capacity > ctx->tstage_cap[ti])
  {
    if (hipSetDevice(ctx->device) != hipSuccess) return nullptr;
    retire_host(ctx, ctx->tstage[ti]);  // (a copy may still be reading it: freed at the next frame's start)
    ctx->tstage[ti] = nullptr;
    ctx->tstage_cap[ti] = 0;
    // the map grows keyframe after keyframe at the start of a sequence: doubling keeps the (slow) pinned
    // re-allocations to a handful
    const int cap = std::max(2 * capacity, 65536);
    if (hipHostMalloc((void**)&ctx->tstage[ti], (size_t)cap * sizeof(lsa_point_t), hipHostMallocDefault) != hipSuccess) return nullptr;
    ctx->tstage_cap[ti] = cap;
  }
  return ctx->tstage[ti];
}

int lsa_set_target_staged(lsa_ctx* ctx, int slot, int type, int m)
{
  if (!ctx || slot < 0 || slot > 1 || type < 0 || type > 2 || m < 0) return ctx ? ctx->fail(LSA_E_ARG, "lsa_set_target_staged: bad argument") : LSA_E_ARG;
  const int ti = slot * 3 + type;
  if (m > ctx->tstage_cap[ti]) return ctx->fail(LSA_E_STATE, "lsa_set_target_staged: more points than the staging buffer holds");
  LSA_HIP(ctx, hipSetDevice(ctx->device));
  if (slot == LSA_TARGET_MAP && ctx->map_ahead_ready[type])
  {
    // uploaded and built ahead on the look-ahead stream (lsa_stage_target_ahead): taken over when it is the same
    // staged cloud with the same cell size; either way its copy out of the staging buffer has to be over
    ctx->map_ahead_ready[type] = false;
    Target& spare = ctx->target[9 + type];
    if (m > 0 && spare.m == m && spare.cell_hint == ctx->target[ti].cell_hint)
    {
      std::swap(ctx->target[ti], spare);
      ctx->target[ti].dirty = false;
      ctx->map_ahead_adopted++;
      LSA_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_map_ahead[type], 0));
      return LSA_OK;
    }
    LSA_HIP(ctx, hipEventSynchronize(ctx->ev_map_ahead[type]));
  }
  int rc = ensure_target(ctx, ti, m);
  if (rc) return rc;
  Target& t = ctx->target[ti];
  t.m = m;
  if (m == 0) return LSA_OK;
  {
    ProfScope ps(ctx, "target_upload_h2d", (double)m * 32);
    LSA_HIP(ctx, hipMemcpyAsync(t.pts, ctx->tstage[ti], (size_t)m * sizeof(lsa_point_t), hipMemcpyHostToDevice, ctx->stream));
  }
  t.dirty = true;  // the search grid is built with the other pending targets at the next match
  return LSA_OK;
}

int lsa_stage_target_ahead(lsa_ctx* ctx, int slot, int type, int m)
{
  if (!ctx || slot != LSA_TARGET_MAP || type < 0 || type > 2 || m < 0) return ctx ? ctx->fail(LSA_E_ARG, "lsa_stage_target_ahead: bad argument") : LSA_E_ARG;
  const int ti = slot * 3 + type;
  if (m > ctx->tstage_cap[ti]) return ctx->fail(LSA_E_STATE, "lsa_stage_target_ahead: more points than the staging buffer holds");
  LSA_HIP(ctx, hipSetDevice(ctx->device));
  if (ctx->map_ahead_ready[type]) LSA_HIP(ctx, hipEventSynchronize(ctx->ev_map_ahead[type]));
  ctx->map_ahead_ready[type] = false;
  if (m == 0) return LSA_OK;
  int rc = ensure_target(ctx, 9 + type, m);
  if (rc) return rc;
  Target& t = ctx->target[9 + type];
  t.m = m;
  t.cell_hint = ctx->target[ti].cell_hint;
  LSA_HIP(ctx, hipMemcpyAsync(t.pts, ctx->tstage[ti], (size_t)m * sizeof(lsa_point_t), hipMemcpyHostToDevice, ctx->prefetch_stream));
  const int tis[1] = {9 + type};
  rc = build_grids(ctx, tis, 1, ctx->prefetch_stream);
  if (rc) return rc;
  LSA_HIP(ctx, hipEventRecord(ctx->ev_map_ahead[type], ctx->prefetch_stream));
  ctx->map_ahead_ready[type] = true;
  return LSA_OK;
}

int lsa_drop_target_ahead(lsa_ctx* ctx, int slot, int type)
{
  if (!ctx || slot != LSA_TARGET_MAP || type < 0 || type > 2) return ctx ? ctx->fail(LSA_E_ARG, "lsa_drop_target_ahead: bad argument") : LSA_E_ARG;
  if (!ctx->map_ahead_ready[type]) return LSA_OK;
  LSA_HIP(ctx, hipSetDevice(ctx->device));
  LSA_HIP(ctx, hipEventSynchronize(ctx->ev_map_ahead[type]));  // the staging buffer is free to be rewritten after this
  ctx->map_ahead_ready[type] = false;
  return LSA_OK;
}

int lsa_staged_targets_adopted(const lsa_ctx* ctx) { return ctx ? ctx->map_ahead_adopted : 0; }

int lsa_mailbox_active(const lsa_ctx* ctx) { return ctx && ctx->mailbox ? 1 : 0; }

int lsa_set_target_from_set(lsa_ctx* ctx, int slot, int type, int set)
{
  if (!ctx || slot < 0 || slot > 1 || type < 0 || type > 2 || set < 0 || set > 2) return ctx ? ctx->fail(LSA_E_ARG, "lsa_set_target_from_set: bad argument") : LSA_E_ARG;
  LSA_HIP(ctx, hipSetDevice(ctx->device));
  const int m = ctx->kp_n[set][type];
  const int ti = slot * 3 + type;
  if (slot == LSA_TARGET_PREVIOUS && set == LSA_SET_RAW_PREVIOUS && ctx->spare_ready[type])
  {
    // built ahead, beside the previous frame's registration (lsa_prepare_previous_targets): taken over if it still
    // describes this very set and was built with the cell size asked for now
    Target& spare = ctx->target[6 + type];
    ctx->spare_ready[type] = false;
    if (m > 0 && spare.m == m && ctx->spare_ver[type] == ctx->kp_ver[set][type] && spare.cell_hint == ctx->target[ti].cell_hint)
    {
      std::swap(ctx->target[ti], spare);
      ctx->target[ti].dirty = false;
      ctx->spare_adopted++;
      LSA_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_spare, 0));
      return LSA_OK;
    }
  }
  int rc = ensure_target(ctx, ti, m);
  if (rc) return rc;
  Target& t = ctx->target[ti];
  t.m = m;
  if (m == 0) return LSA_OK;
  LSA_HIP(ctx, hipMemcpyAsync(t.pts, ctx->kp[set][type], (size_t)m * sizeof(lsa_point_t), hipMemcpyDeviceToDevice, ctx->stream));
  t.dirty = true;
  return LSA_OK;
}

int lsa_prepare_previous_targets(lsa_ctx* ctx, unsigned type_mask)
{
  if (!ctx || (type_mask & ~7u)) return ctx ? ctx->fail(LSA_E_ARG, "lsa_prepare_previous_targets: bad argument") : LSA_E_ARG;
  LSA_HIP(ctx, hipSetDevice(ctx->device));
  int tis[3], n = 0;
  for (int k = 0; k < 3; ++k)
  {
    ctx->spare_ready[k] = false;
    const int m = ctx->kp_n[LSA_SET_RAW_CURRENT][k];
    if (!((type_mask >> k) & 1u) || m <= 0) continue;
    int rc = ensure_target(ctx, 6 + k, m);
    if (rc) return rc;
    Target& t = ctx->target[6 + k];
    t.m = m;
    t.cell_hint = ctx->target[LSA_TARGET_PREVIOUS * 3 + k].cell_hint;
    tis[n++] = 6 + k;
  }
  if (n == 0) return LSA_OK;
  // the keypoints are final once everything enqueued so far has run; the copies and the grid build follow on the
  // look-ahead stream, beside whatever comes next on the context's stream
  LSA_HIP(ctx, hipEventRecord(ctx->ev_kp_ready, ctx->stream));
  LSA_HIP(ctx, hipStreamWaitEvent(ctx->prefetch_stream, ctx->ev_kp_ready, 0));
  for (int i = 0; i < n; ++i)
  {
    const int k = tis[i] - 6;
    LSA_HIP(ctx, hipMemcpyAsync(ctx->target[tis[i]].pts, ctx->kp[LSA_SET_RAW_CURRENT][k], (size_t)ctx->target[tis[i]].m * sizeof(lsa_point_t),
                                hipMemcpyDeviceToDevice, ctx->prefetch_stream));
  }
  int rc = build_grids(ctx, tis, n, ctx->prefetch_stream);
  if (rc) return rc;
  LSA_HIP(ctx, hipEventRecord(ctx->ev_spare, ctx->prefetch_stream));
  for (int i = 0; i < n; ++i)
  {
    const int k = tis[i] - 6;
    ctx->spare_ver[k] = ctx->kp_ver[LSA_SET_RAW_CURRENT][k];
    ctx->spare_ready[k] = true;
  }
  return LSA_OK;
}

int lsa_prepared_targets_adopted(const lsa_ctx* ctx) { return ctx ? ctx->spare_adopted : 0; }

int lsa_set_knn_rounds(lsa_ctx* ctx, int type, int rounds)
{
  if (!ctx || type < 0 || type > 2 || rounds < 2 || rounds > 3) return LSA_E_ARG;
  ctx->knn_rounds[type] = rounds;
  return LSA_OK;
}

int lsa_set_fused_match(lsa_ctx* ctx, int on)
{
  if (!ctx) return LSA_E_ARG;
  ctx->fused_match = on != 0;
  ctx->fused_model = on != 2;
  return LSA_OK;
}

int lsa_set_knn_lanes(lsa_ctx* ctx, int type, int lanes)
{
  if (!ctx || type < 0 || type > 2 || lanes < 1) return LSA_E_ARG;
  ctx->knn_lanes[type] = lanes;
  return LSA_OK;
}

int lsa_download_target(lsa_ctx* ctx, int slot, int type, lsa_point_t* out, int capacity)
{
  if (!ctx || slot < 0 || slot > 1 || type < 0 || type > 2 || !out) return ctx ? ctx->fail(LSA_E_ARG, "lsa_download_target: bad argument") : LSA_E_ARG;
  const Target& t = ctx->target[slot * 3 + type];
  const int n = std::min(capacity, t.m);
  if (n <= 0) return 0;
  LSA_HIP(ctx, hipSetDevice(ctx->device));
  LSA_HIP(ctx, hipMemcpyAsync(out, t.pts, (size_t)n * sizeof(lsa_point_t), hipMemcpyDeviceToHost, ctx->stream));
  LSA_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return n;
}

int lsa_target_size(const lsa_ctx* ctx, int slot, int type) { return (ctx && slot >= 0 && slot <= 1 && type >= 0 && type <= 2) ? ctx->target[slot * 3 + type].m : LSA_E_ARG; }

int lsa_set_target_cell_size(lsa_ctx* ctx, int slot, int type, float cell)
{
  if (!ctx || slot < 0 || slot > 1 || type < 0 || type > 2 || !(cell > 0.f)) return LSA_E_ARG;
  ctx->target[slot * 3 + type].cell_hint = cell;
  return LSA_OK;
}

// Next block of the type's histogram ring.  The ring is cleared one half at a time, when the position enters the
// half: everything that used those blocks finished long ago (the streams were joined and the host has read the
// results of those matches since), and the blocks of the last kHistRing / 2 matches stay readable
// (lsa_match_histogram).
static int next_histogram(lsa_ctx* ctx, int type, hipStream_t st, int** hist)
{
  constexpr int half = kHistRing / 2;
  ++ctx->hist_serial[type];
  if (++ctx->hist_pos[type] >= kHistRing) ctx->hist_pos[type] = 0;
  const int pos = ctx->hist_pos[type];
  if (pos % half == 0)
    LSA_HIP(ctx, hipMemsetAsync(ctx->hist_dev + ((size_t)type * kHistRing + pos) * 16, 0, (size_t)half * 16 * sizeof(int), st));
  *hist = ctx->hist_dev + ((size_t)type * kHistRing + pos) * 16;
  return LSA_OK;
}

// Resolves the parameters of one keypoint type's match and takes its histogram block.  Returns 1 when there is
// something to search (keypoints and a non-empty target), 0 when the match is complete as it stands.
static int match_prepare(lsa_ctx* ctx, int slot, int type, int query_set, const lsa_match_params_t* p, hipStream_t st, MatchPrep& out)
{
  const int nq = ctx->kp_n[query_set][type];
  MatchBuf& mb = ctx->match[type];
  int* hist = nullptr;
  {
    const int rc = next_histogram(ctx, type, st, &hist);
    if (rc) return rc;
  }
  mb.k = nq;
  mb.sat = p->saturation_distance;
  mb.valid = true;
  ctx->last_match_type = type;
  if (nq == 0) return 0;
  const int ti = slot * 3 + type;
  Target& t = ctx->target[ti];
  if (t.m == 0)
  {
    // empty target: MatchingResults::Reset leaves every keypoint UNKOWN and the histogram empty
    // (KeypointsMatcher.cxx:53-58)
    hipLaunchKernelGGL(k_fill_status, dim3((nq + 255) / 256), dim3(256), 0, st, mb.status, mb.rec, mb.cap, nq, (uint8_t)LSA_MATCH_UNKOWN);
    return 0;
  }
  MatchConst mc;
  mc.type = type;
  mc.single_edge_per_ring = p->single_edge_per_ring;
  mc.min_neighbors = p->edge_min_nb_neighbors;
  mc.max_dist2 = p->max_neighbors_distance * p->max_neighbors_distance;
  mc.planarity = p->planarity_threshold;
  mc.bad_param = 0;
  const double e2 = p->edge_max_model_error * p->edge_max_model_error;
  mc.ransac_sq_inlier = (float)e2;
  if (type == LSA_EDGE)
  {
    mc.k = p->edge_nb_neighbors;
    mc.max_model_err = p->edge_max_model_error;
    if (p->edge_nb_neighbors < 2 || p->edge_min_nb_neighbors < 2) mc.bad_param = 1;
  }
  else if (type == LSA_PLANE)
  {
    mc.k = p->plane_nb_neighbors;
    mc.max_model_err = p->plane_max_model_error;
    if (p->plane_nb_neighbors < 3) mc.bad_param = 1;
  }
  else
  {
    mc.k = p->blob_nb_neighbors;
    mc.max_model_err = 0.;
    if (p->blob_nb_neighbors < 4) mc.bad_param = 1;
  }
  if (mc.k < 1) mc.k = 1;
  // planes and blobs use all k neighbours and reject the match when the k-th is too far: the search may
  // stop as soon as that is certain (and the target is known to hold at least k points).  Edge matches
  // filter their neighbours first, so they need the true k nearest whatever the distance.
  float far_d2 = INFINITY;
  if (type != LSA_EDGE && t.m >= mc.k) far_d2 = (float)(mc.max_dist2 * 1.0001);
  out.type = type;
  out.ti = ti;
  out.queries = ctx->kp[query_set][type];
  out.nq = nq;
  out.mc = mc;
  out.far_d2 = far_d2;
  out.hist = hist;
  return 1;
}

// Enqueues one prepared match as the staged kernels (first kNN stage -> second stage -> model fit) on `st`.
static void match_enqueue_staged(lsa_ctx* ctx, const MatchPrep& mp, const double pose[16], hipStream_t st)
{
  const int type = mp.type, ti = mp.ti, nq = mp.nq;
  const MatchConst& mc = mp.mc;
  const lsa_point_t* q = mp.queries;
  int* hist = mp.hist;
  Rigid rp;
  row_major_to_rt(pose, rp.R, rp.t);
  if (!mc.bad_param)
  {
    if (mc.k <= 5) launch_knn<5>(ctx, q, nq, rp, mc.k, mp.far_d2, type, ti, st, hist);  // planes: 5 neighbours, fewer registers, more waves in flight
    else if (mc.k <= 8) launch_knn<8>(ctx, q, nq, rp, mc.k, mp.far_d2, type, ti, st, hist);
    else launch_knn<16>(ctx, q, nq, rp, mc.k, mp.far_d2, type, ti, st, hist);
  }
  {
    ProfScope ps(ctx, type == LSA_EDGE ? "model_edge" : type == LSA_PLANE ? "model_plane" : "model_blob",
                 (double)nq * (32 + mc.k * 8 + mc.k * 16 + 136), st);
    if (type == LSA_EDGE)
    {
      if (mc.k <= 8) launch_model<8, LSA_EDGE>(ctx, q, nq, mc, type, ti, st, hist);
      else launch_model<16, LSA_EDGE>(ctx, q, nq, mc, type, ti, st, hist);
    }
    else if (type == LSA_PLANE)
    {
      if (mc.k <= 8) launch_model<8, LSA_PLANE>(ctx, q, nq, mc, type, ti, st, hist);
      else launch_model<16, LSA_PLANE>(ctx, q, nq, mc, type, ti, st, hist);
    }
    else
    {
      if (mc.k <= 8) launch_model<8, LSA_BLOB>(ctx, q, nq, mc, type, ti, st, hist);
      else launch_model<16, LSA_BLOB>(ctx, q, nq, mc, type, ti, st, hist);
    }
  }
}

static int match_check(lsa_ctx* ctx, int type, const lsa_match_params_t* p)
{
  const int k = type == LSA_EDGE ? p->edge_nb_neighbors : type == LSA_PLANE ? p->plane_nb_neighbors : p->blob_nb_neighbors;
  if (k > kKnnMax) return ctx->fail(LSA_E_ARG, "lsa_match: more than 16 neighbours requested");
  return LSA_OK;
}

int lsa_match(lsa_ctx* ctx, int slot, int type, int query_set, const lsa_match_params_t* p, const double pose[16], int histogram[LSA_MATCH_NSTATUS])
{
  if (!ctx || !p || !pose || slot < 0 || slot > 1 || type < 0 || type > 2 || query_set < 0 || query_set > 2)
    return ctx ? ctx->fail(LSA_E_ARG, "lsa_match: bad argument") : LSA_E_ARG;
  int hist3[3 * LSA_MATCH_NSTATUS];
  const int rc = lsa_match_types(ctx, slot, 1u << type, query_set, p, pose, histogram ? hist3 : nullptr);
  if (rc == LSA_OK && histogram) std::memcpy(histogram, hist3 + type * LSA_MATCH_NSTATUS, LSA_MATCH_NSTATUS * sizeof(int));
  return rc;
}

static int match_types_impl(lsa_ctx* ctx, int slot, unsigned type_mask, int query_set, const lsa_match_params_t* p, const double pose[16], int* histograms,
                            const InterpConst* undistort, int gate = -1, bool gate_undistorts = false)
{
  if (!ctx || !p || (!pose && gate < 0) || slot < 0 || slot > 1 || (type_mask & ~7u) || query_set < 0 || query_set > 2)
    return ctx ? ctx->fail(LSA_E_ARG, "lsa_match_types: bad argument") : LSA_E_ARG;
  LSA_HIP(ctx, hipSetDevice(ctx->device));
  if (histograms) std::memset(histograms, 0, 3 * LSA_MATCH_NSTATUS * sizeof(int));
  int types[3], nt = 0;
  for (int k = 0; k < 3; ++k)
    if ((type_mask >> k) & 1u)
    {
      int rc = match_check(ctx, k, p);
      if (rc) return rc;
      rc = ensure_match(ctx, k, ctx->kp_n[query_set][k]);  // may reallocate (and synchronise) before anything is forked
      if (rc) return rc;
      types[nt++] = k;
    }
  if (nt == 0) return LSA_OK;
  {
    const int rc = flush_grids(ctx);
    if (rc) return rc;
  }
  if (ctx->fused_match)
  {
    // one launch for all the types (+ the tail launch), on the context's stream
    MatchPrep preps[3];
    int np = 0;
    for (int i = 0; i < nt; ++i)
    {
      if (gate >= 0)
      {
        // should the iteration be called off, what it announces here is taken back (lsa_icp_cancel)
        lsa_ctx::GateSaved& sv = ctx->gate_saved[gate];
        const int k = types[i];
        sv.mask |= 1u << k;
        sv.sat[k] = ctx->match[k].sat; sv.k[k] = ctx->match[k].k; sv.valid[k] = ctx->match[k].valid;
        sv.hist_pos[k] = ctx->hist_pos[k]; sv.hist_serial[k] = ctx->hist_serial[k];
      }
      const int rc = match_prepare(ctx, slot, types[i], query_set, p, ctx->stream, preps[np]);
      if (rc < 0) return rc;
      if (rc > 0) ++np;
    }
    if (np > 0)
    {
      const int rc = enqueue_fused_match(ctx, preps, np, pose, ctx->stream, undistort, gate, gate_undistorts);
      if (rc) return rc;
    }
  }
  else
  {
    // fork: the first type stays on the context stream, the others run beside it.  A single type's kernels
    // leave most of the 256 CUs idle (a few thousand queries, latency bound), so the types overlap almost fully.
    for (int i = 0; i + 1 < nt; ++i)
      if (!ctx->side_stream[i]) LSA_HIP(ctx, hipStreamCreateWithFlags(&ctx->side_stream[i], hipStreamNonBlocking));  // lsa_ctx_create: why not there
    if (nt > 1) LSA_HIP(ctx, hipEventRecord(ctx->ev_fork, ctx->stream));
    for (int i = 0; i < nt; ++i)
    {
      hipStream_t st = i == 0 ? ctx->stream : ctx->side_stream[i - 1];
      if (i > 0) LSA_HIP(ctx, hipStreamWaitEvent(st, ctx->ev_fork, 0));
      MatchPrep mp;
      const int rc = match_prepare(ctx, slot, types[i], query_set, p, st, mp);
      if (rc < 0) return rc;
      if (rc > 0) match_enqueue_staged(ctx, mp, pose, st);
      if (i > 0) LSA_HIP(ctx, hipEventRecord(ctx->ev_join[i - 1], st));
    }
    for (int i = 1; i < nt; ++i) LSA_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_join[i - 1], 0));
  }
  if (histograms)
  {
    int* hp = reinterpret_cast<int*>(ctx->host_pinned) + 32;
    for (int i = 0; i < nt; ++i)
      LSA_HIP(ctx, hipMemcpyAsync(hp + types[i] * 16, ctx->hist_dev + ((size_t)types[i] * kHistRing + ctx->hist_pos[types[i]]) * 16, 16 * sizeof(int),
                                  hipMemcpyDeviceToHost, ctx->stream));
    LSA_HIP(ctx, hipStreamSynchronize(ctx->stream));
    for (int i = 0; i < nt; ++i)
      for (int s = 0; s < LSA_MATCH_NSTATUS; ++s) histograms[types[i] * LSA_MATCH_NSTATUS + s] = hp[types[i] * 16 + s];
  }
  return LSA_OK;
}

int lsa_match_types(lsa_ctx* ctx, int slot, unsigned type_mask, int query_set, const lsa_match_params_t* p, const double pose[16], int* histograms)
{
  return match_types_impl(ctx, slot, type_mask, query_set, p, pose, histograms, nullptr);
}

int lsa_match_types_undistorted(lsa_ctx* ctx, int slot, unsigned type_mask, const lsa_match_params_t* p, const double pose[16], int* histograms, const double H0[16],
                                const double H1[16], double t0, double t1)
{
  if (!ctx || !p || !pose || !H0 || !H1 || slot < 0 || slot > 1 || (type_mask & ~7u))
    return ctx ? ctx->fail(LSA_E_ARG, "lsa_match_types_undistorted: bad argument") : LSA_E_ARG;
  // one launch when the search kernel reaches every keypoint of the working set: the one-launch form, every type that has
  // keypoints asked for, with a target and with valid parameters (a type that is not searched would keep its distortion)
  bool reach = ctx->fused_match;
  for (int k = 0; k < 3 && reach; ++k)
  {
    if (ctx->kp_n[LSA_SET_WORKING][k] <= 0) continue;
    const int nb = k == LSA_EDGE ? p->edge_nb_neighbors : k == LSA_PLANE ? p->plane_nb_neighbors : p->blob_nb_neighbors;
    const bool bad = k == LSA_EDGE ? (nb < 2 || p->edge_min_nb_neighbors < 2) : k == LSA_PLANE ? nb < 3 : nb < 4;
    reach = ((type_mask >> k) & 1u) && ctx->target[slot * 3 + k].m > 0 && !bad && nb <= kKnnMax;
  }
  if (!reach)
  {
    const int rc = lsa_undistort(ctx, H0, H1, t0, t1);
    return rc ? rc : match_types_impl(ctx, slot, type_mask, LSA_SET_WORKING, p, pose, histograms, nullptr);
  }
  const InterpConst ic = make_interp_const(H0, H1, t0, t1);
  return match_types_impl(ctx, slot, type_mask, LSA_SET_WORKING, p, pose, histograms, &ic);
}

// Whether lsa_match_types_undistorted would do the undistortion inside the search kernel: the one-launch form, every type
// that has keypoints asked for, with a target and with valid parameters
static bool search_reaches_every_keypoint(lsa_ctx* ctx, int slot, unsigned type_mask, const lsa_match_params_t* p)
{
  bool reach = ctx->fused_match;
  for (int k = 0; k < 3 && reach; ++k)
  {
    if (ctx->kp_n[LSA_SET_WORKING][k] <= 0) continue;
    const int nb = k == LSA_EDGE ? p->edge_nb_neighbors : k == LSA_PLANE ? p->plane_nb_neighbors : p->blob_nb_neighbors;
    const bool bad = k == LSA_EDGE ? (nb < 2 || p->edge_min_nb_neighbors < 2) : k == LSA_PLANE ? nb < 3 : nb < 4;
    reach = ((type_mask >> k) & 1u) && ctx->target[slot * 3 + k].m > 0 && !bad && nb <= kKnnMax;
  }
  return reach;
}

int lsa_match_types_gated(lsa_ctx* ctx, int slot, unsigned type_mask, int query_set, const lsa_match_params_t* p, int undistort)
{
  if (!ctx || !p || slot < 0 || slot > 1 || (type_mask & ~7u) || query_set < 0 || query_set > 2)
    return ctx ? ctx->fail(LSA_E_ARG, "lsa_match_types_gated: bad argument") : LSA_E_ARG;
  if (ctx->gate_current < 0) return ctx->fail(LSA_E_STATE, "lsa_match_types_gated: no gate to wait behind (lsa_icp_gate)");
  // only the one-launch form reads a gate; an undistortion must reach every keypoint of the working set (otherwise it is a
  // launch of its own, which the caller has to enqueue once the motion is known)
  if (!ctx->fused_match || !ctx->fused_model) return 1;
  if (undistort && (query_set != LSA_SET_WORKING || !search_reaches_every_keypoint(ctx, slot, type_mask, p))) return 1;
  for (int k = 0; k < 3; ++k)
    if (((type_mask >> k) & 1u) && ctx->kp_n[query_set][k] > 0 && ctx->target[slot * 3 + k].m <= 0) return 1;  // (an empty target is answered by a fill, not by the search)
  return match_types_impl(ctx, slot, type_mask, query_set, p, nullptr, nullptr, nullptr, ctx->gate_current, undistort != 0);
}

int lsa_overlap(lsa_ctx* ctx, unsigned type_mask, int interpolate, const double H0[16], const double H1[16], double t0, double t1, float sampling_ratio,
                const double leaf_size[3], float* overlap)
{
  if (!ctx || !H0 || (interpolate && !H1) || !leaf_size || !overlap || (type_mask & ~7u))
    return ctx ? ctx->fail(LSA_E_ARG, "lsa_overlap: bad argument") : LSA_E_ARG;
  if (!ctx->frame || ctx->frame_n <= 0) return ctx->fail(LSA_E_STATE, "lsa_overlap: no frame");
  LSA_HIP(ctx, hipSetDevice(ctx->device));
  *overlap = -1.f;
  const int nb = (int)(ctx->frame_n * sampling_ratio);  // size_t * float -> float -> int (ConfidenceEstimators.cxx:33)
  unsigned used = 0;
  for (int k = 0; k < 3; ++k)
    if (((type_mask >> k) & 1u) && ctx->target[LSA_TARGET_MAP * 3 + k].m > 0) used |= 1u << k;
  if (nb <= 0 || used == 0) return LSA_OK;
  if (nb > ctx->frame_n) return ctx->fail(LSA_E_ARG, "lsa_overlap: sampling ratio above 1");
  int rc = ensure_scratch(ctx, (size_t)nb * sizeof(lsa_point_t) + 1024 * sizeof(float));
  if (rc) return rc;
  for (int k = 0; k < 3; ++k)
    if ((used >> k) & 1u)
    {
      MatchBuf& mb = ctx->match[k];
      const int cap0 = mb.cap;
      rc = ensure_match(ctx, k, nb);
      if (rc) return rc;
      if (mb.cap != cap0) mb.valid = false;  // the records moved with the buffers
    }
  rc = flush_grids(ctx);
  if (rc) return rc;
  hipStream_t st = ctx->stream;
  float4* q4 = reinterpret_cast<float4*>(ctx->scratch_out);
  float* partials = reinterpret_cast<float*>(reinterpret_cast<char*>(ctx->scratch_out) + (size_t)nb * sizeof(lsa_point_t));
  Rigid T;
  row_major_to_rt(H0, T.R, T.t);
  InterpConst ic{};
  if (interpolate) ic = make_interp_const(H0, H1, t0, t1);
  Rigid ident;
  for (int i = 0; i < 9; ++i) ident.R[i] = (i % 4 == 0) ? 1. : 0.;
  ident.t[0] = ident.t[1] = ident.t[2] = 0.;
  OverlapConst oc;
  {
    ProfScope ps(ctx, "overlap_lcp", (double)nb * (32 + 3 * (32 + 24) + 12));
    hipLaunchKernelGGL(k_overlap_queries, dim3((nb + 255) / 256), dim3(256), 0, st, reinterpret_cast<const float4*>(ctx->frame), nb, sampling_ratio,
                       interpolate, ic, T, q4);
    for (int k = 0; k < 3; ++k)
    {
      oc.d2[k] = nullptr;
      oc.inv2sq[k] = 0.f;
      if (!((used >> k) & 1u)) continue;
      int* hist = nullptr;
      {
        const int rc = next_histogram(ctx, k, st, &hist);
        if (rc) return rc;
      }
      // nearest neighbour = the k = 1 case of the exact search (identity pose: the queries are world points already)
      launch_knn<5>(ctx, reinterpret_cast<const lsa_point_t*>(q4), nb, ident, 1, INFINITY, k, LSA_TARGET_MAP * 3 + k, st, hist);
      oc.d2[k] = ctx->match[k].knn_d2;
      const float sq = (float)std::pow(leaf_size[k] / 3.f, 2);  // std::pow(GetLeafSize() / 3.f, 2) (:55)
      oc.inv2sq[k] = 1.f / (2.f * sq);
    }
    hipLaunchKernelGGL(k_overlap_score, dim3(256), dim3(256), 0, st, oc, nb, partials);
  }
  float* hp = reinterpret_cast<float*>(ctx->host_pinned + 192);
  LSA_HIP(ctx, hipMemcpyAsync(hp, partials, 256 * sizeof(float), hipMemcpyDeviceToHost, st));
  LSA_HIP(ctx, hipStreamSynchronize(st));
  float lcp = 0.f;
  for (int b = 0; b < 256; ++b) lcp += hp[b];
  *overlap = lcp / nb;
  return LSA_OK;
}

int lsa_download_match(lsa_ctx* ctx, int type, uint8_t* status, double* weights, double* records, int capacity)
{
  if (!ctx || type < 0 || type > 2 || !status || !weights) return ctx ? ctx->fail(LSA_E_ARG, "lsa_download_match: bad argument") : LSA_E_ARG;
  MatchBuf& mb = ctx->match[type];
  if (!mb.valid) return 0;
  const int n = std::min(capacity, mb.k);
  if (n <= 0) return 0;
  const size_t bytes = (size_t)n * (16 + 1) * sizeof(double);
  int rc = ensure_scratch(ctx, bytes);
  if (rc) return rc;
  double* drec = (double*)ctx->scratch_out;
  double* dw = drec + (size_t)n * 16;
  hipLaunchKernelGGL(k_records_to_aos, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, mb.rec, mb.status, mb.cap, n, records ? drec : nullptr, dw);
  if (records) LSA_HIP(ctx, hipMemcpyAsync(records, drec, (size_t)n * 16 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  LSA_HIP(ctx, hipMemcpyAsync(weights, dw, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  LSA_HIP(ctx, hipMemcpyAsync(status, mb.status, (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
  LSA_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return n;
}

long long lsa_match_serial(const lsa_ctx* ctx, int type)
{
  if (!ctx || type < 0 || type > 2) return LSA_E_ARG;
  return ctx->hist_serial[type];
}

int lsa_match_histogram(lsa_ctx* ctx, int type, long long serial, int histogram[LSA_MATCH_NSTATUS])
{
  if (!ctx || type < 0 || type > 2 || !histogram) return ctx ? ctx->fail(LSA_E_ARG, "lsa_match_histogram: bad argument") : LSA_E_ARG;
  const long long back = ctx->hist_serial[type] - serial;
  if (serial <= 0 || back < 0 || back >= kHistRing / 2) return ctx->fail(LSA_E_STATE, "lsa_match_histogram: that match is not (or no longer) in the ring");
  const int pos = (int)(((long long)ctx->hist_pos[type] - back) % kHistRing + kHistRing) % kHistRing;
  LSA_HIP(ctx, hipSetDevice(ctx->device));
  LSA_HIP(ctx, hipStreamSynchronize(ctx->stream));
  LSA_HIP(ctx, hipMemcpy(histogram, ctx->hist_dev + ((size_t)type * kHistRing + pos) * 16, LSA_MATCH_NSTATUS * sizeof(int), hipMemcpyDeviceToHost));
  return LSA_OK;
}

int lsa_match_slow_queries(lsa_ctx* ctx)
{
  if (!ctx) return LSA_E_ARG;
  int v = 0;
  if (hipStreamSynchronize(ctx->stream) != hipSuccess) return LSA_E_HIP;
  if (hipMemcpy(&v, ctx->hist_dev + ((size_t)ctx->last_match_type * kHistRing + ctx->hist_pos[ctx->last_match_type]) * 16 + LSA_MATCH_NSTATUS, sizeof(int), hipMemcpyDeviceToHost) != hipSuccess) return LSA_E_HIP;
  return v;
}

int lsa_match_route_stats(lsa_ctx* ctx, int type, int out[8])
{
  if (!ctx || type < 0 || type > 2 || !out) return LSA_E_ARG;
  if (hipStreamSynchronize(ctx->stream) != hipSuccess) return LSA_E_HIP;
  if (hipMemcpy(out, ctx->hist_dev + ((size_t)type * kHistRing + ctx->hist_pos[type]) * 16 + LSA_MATCH_NSTATUS, 8 * sizeof(int), hipMemcpyDeviceToHost) != hipSuccess) return LSA_E_HIP;
  return LSA_OK;
}

int lsa_match_trace(lsa_ctx* ctx, unsigned long long* out, int blocks)
{
  if (!ctx || !out || blocks < 0 || blocks > 8192 || !ctx->trace_dev) return LSA_E_ARG;
  if (hipStreamSynchronize(ctx->stream) != hipSuccess) return LSA_E_HIP;
  if (hipMemcpy(out, ctx->trace_dev, (size_t)blocks * 12 * sizeof(unsigned long long), hipMemcpyDeviceToHost) != hipSuccess) return LSA_E_HIP;
  return LSA_OK;
}

int lsa_match_exhaustive_queries(lsa_ctx* ctx)
{
  if (!ctx) return LSA_E_ARG;
  int v = 0;
  if (hipStreamSynchronize(ctx->stream) != hipSuccess) return LSA_E_HIP;
  if (hipMemcpy(&v, ctx->hist_dev + ((size_t)ctx->last_match_type * kHistRing + ctx->hist_pos[ctx->last_match_type]) * 16 + LSA_MATCH_NSTATUS + 1, sizeof(int), hipMemcpyDeviceToHost) != hipSuccess) return LSA_E_HIP;
  return v;
}

int lsa_accumulate(lsa_ctx* ctx, unsigned type_mask, const double w[6], int want_jacobian, double* cost, double g[6], double H[36], int* n_valid)
{
  if (!ctx || !w || !cost) return ctx ? ctx->fail(LSA_E_ARG, "lsa_accumulate: bad argument") : LSA_E_ARG;
  LSA_HIP(ctx, hipSetDevice(ctx->device));
  AccumConst c;
  // R = Rz Ry Rx and its partial derivatives (CeresCostFunctions.h:67-79), once per evaluation on the host
  rotation_and_derivatives(lsa_cos(w[3]), lsa_sin(w[3]), lsa_cos(w[4]), lsa_sin(w[4]), lsa_cos(w[5]), lsa_sin(w[5]), c.rot.R, c.rot.dRx, c.rot.dRy, c.rot.dRz);
  c.rot.t[0] = w[0]; c.rot.t[1] = w[1]; c.rot.t[2] = w[2];
  int total = 0;
  for (int k = 0; k < 3; ++k)
  {
    MatchBuf& mb = ctx->match[k];
    const bool use = (type_mask >> k) & 1u && mb.valid && mb.k > 0;
    c.set.rec[k] = mb.rec; c.set.status[k] = mb.status; c.set.cap[k] = mb.cap;
    c.set.count[k] = use ? mb.k : 0;
    c.set.sat2[k] = mb.sat * mb.sat;
    total += c.set.count[k];
  }
  c.jac = want_jacobian;
  hipStream_t st = ctx->stream;
  const unsigned want = (unsigned)(++ctx->mailbox_seq);
  {
    ProfScope ps(ctx, want_jacobian ? "accumulate_jac" : "accumulate_cost", (double)total * 129);
    hipLaunchKernelGGL(k_accumulate, dim3(ctx->accum_blocks), dim3(256), 0, st, c, ctx->partials, ctx->mailbox, want);
  }
  double* hp = ctx->host_pinned + 64;
  bool got = false;
  if (ctx->mailbox)
  {
    // poll the granules (bounded: fall back to a device fold + synchronous copy if one does not arrive); the blocks
    // are folded in index order as they come in
    const auto t0 = std::chrono::steady_clock::now();
    unsigned spins = 0;
    bool timeout = false;
    for (int v = 0; v < kAccumVals; ++v) hp[v] = 0.;
    for (int b = 0; b < ctx->accum_blocks && !timeout; ++b)
    {
      const unsigned long long* row = ctx->mailbox + (size_t)b * kMailboxStride;
      for (int v = 0; v < kAccumVals && !timeout; ++v)
      {
        unsigned long long g[2];
        for (int h = 0; h < 2 && !timeout; ++h)
          while (((g[h] = __atomic_load_n(row + 2 * v + h, __ATOMIC_RELAXED)) >> 32) != want)
          {
            if ((++spins & 0x3ff) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(200)) { timeout = true; break; }
#if defined(__x86_64__)
            __builtin_ia32_pause();
#endif
          }
        const unsigned long long bits = ((g[1] & 0xffffffffull) << 32) | (g[0] & 0xffffffffull);
        double d;
        std::memcpy(&d, &bits, sizeof(d));
        hp[v] += d;
      }
    }
    got = !timeout;
  }
  if (!got || ctx->mailbox_check)
  {
    double* dst = got ? ctx->host_pinned + 96 : hp;
    hipLaunchKernelGGL(k_accumulate_final, dim3(1), dim3(64), 0, st, ctx->partials, ctx->accum_blocks, ctx->reduce_out);
    LSA_HIP(ctx, hipMemcpyAsync(dst, ctx->reduce_out, kAccumVals * sizeof(double), hipMemcpyDeviceToHost, st));
    LSA_HIP(ctx, hipStreamSynchronize(st));
    // LSA_MAILBOX_CHECK: what came through the mailbox must be bit for bit what the device folds from its own partials
    if (got && std::memcmp(dst, hp, kAccumVals * sizeof(double)) != 0) return ctx->fail(LSA_E_STATE, "lsa_accumulate: mailbox and device fold disagree");
  }
  *cost = hp[0];
  if (n_valid) *n_valid = (int)hp[28];
  if (g) for (int a = 0; a < 6; ++a) g[a] = hp[1 + a];
  if (H)
  {
    int h = 7;
    for (int a = 0; a < 6; ++a)
      for (int b = a; b < 6; ++b) { H[a * 6 + b] = hp[h]; H[b * 6 + a] = hp[h]; ++h; }
  }
  return LSA_OK;
}

}  // extern "C"
