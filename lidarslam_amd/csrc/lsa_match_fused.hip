// lsa_match_fused.hip -- one ICP iteration's matching step (the `for (auto k : KeypointTypes)` loops of
// slam_lib/src/Slam.cxx:895-912 and 1074-1091 around KeypointsMatcher::BuildMatchResiduals,
// slam_lib/src/KeypointsMatcher.cxx:33-74) as ONE launch for all keypoint types: exact kNN
// (KDTreePCLAdaptor::KnnSearch, slam_lib/include/LidarSlam/KDTreePCLAdaptor.h:79-105) and the model fit
// (KeypointsMatcher.cxx:106-346) of a keypoint in the same kernel, the neighbour lists never leave the chip.
//
// Search.  The target is indexed by a dense grid at three resolutions (cell, 4 x, 16 x; lsa_match.hip builds it).
// A spinning-LiDAR cloud spans three orders of magnitude in density, so no single block size suits every query:
// G lanes share a query and pick the block to search from the CELL COUNTS alone -- the blocks
//   shell 0..6 = (level 0, 3^3 cells) (0, 5^3) (1, 3^3) (1, 5^3) (2, 3^3) (2, 5^3) (2, 7^3)
// in order of the radius they prove; the row bounds of the first three are fetched in one memory round trip.  The
// first block that holds k points is scanned: every lane walks its own rows of the block (a row of cells is one
// contiguous run of the cell-sorted points) and keeps the k best it sees in a sorted list of 64-bit (distance,
// index) keys in registers -- one load in flight ahead of the one being compared, no cross-lane traffic in the
// loop; the G lists are then merged by k rounds of group minimum (DPP).  If the k-th best lies inside the radius
// the block proves, the query is settled; otherwise that distance is an upper bound and the first block proving
// more than it settles the query for certain with one more scan.  Planes and blobs stop as soon as fewer than k
// points lie within a radius beyond MaxNeighborsDistance (NEIGHBORS_TOO_FAR whatever they are).  What even the
// largest block cannot settle (isolated edge keypoints tens of metres from any target point) goes through a
// device list to the tail kernel: one wavefront per query over the whole target.  Same (distance, index) total
// order everywhere => the result does not depend on the route taken.
//
// Model.  The leaders of a block's groups leave their neighbour lists in LDS; after one barrier the first 256 / G
// threads fit the models of the block's keypoints, one thread each, dense -- the other wavefronts are gone and
// their SIMDs take the next block's search.
#include <cmath>
#include "lsa_knn.h"
#include "lsa_match_internal.h"

using namespace lsa;

namespace
{

constexpr int kSlots = 13;    // row slots per lane: 3 + 7 + 3 rows of shells 0..2 at G = 4, 49 rows of a 7^3 block over 4 lanes
constexpr int kPending = -2;  // neighbour count of a query handed to the tail kernel
constexpr int kShells = 7;

__device__ __forceinline__ constexpr int shell_level(int s) { return s < 2 ? 0 : (s < 4 ? 1 : 2); }
__device__ __forceinline__ constexpr int shell_r(int s) { return s == 6 ? 3 : ((s & 1) ? 2 : 1); }

// sum over the G lanes of a group, result in every lane (DPP inside a row of 16 lanes, as group_min)
template <int G>
__device__ __forceinline__ unsigned group_sum(unsigned v)
{
  if (G >= 2) v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, true);
  if (G >= 4) v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, true);
  if (G >= 8) v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xF, 0xF, true);
  if (G >= 16) v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x140, 0xF, 0xF, true);
  if (G >= 32) v += (unsigned)__shfl_xor((int)v, 16);
  if (G >= 64) v += (unsigned)__shfl_xor((int)v, 32);
  return v;
}

// the k best a lane has seen, ascending, in registers; indices are compile-time constants everywhere
template <int KMAX>
struct LaneList
{
  knn_key v[KMAX];
  __device__ __forceinline__ void reset()
  {
#pragma unroll
    for (int s = 0; s < KMAX; ++s) v[s] = kKeyEmpty;
  }
  __device__ __forceinline__ void insert(knn_key key)
  {
    if (key < v[KMAX - 1])
    {
      v[KMAX - 1] = key;
#pragma unroll
      for (int s = KMAX - 1; s > 0; --s)
      {
        const knn_key a = v[s - 1], b = v[s];
        const bool sw = b < a;
        v[s - 1] = sw ? b : a;
        v[s] = sw ? a : b;
      }
    }
  }
};

// best <- the k smallest keys of the G lanes' lists, uniform across the group (the lists are consumed)
template <int KMAX, int G>
__device__ __forceinline__ void merge_lists(LaneList<KMAX>& L, int k, knn_key (&best)[KMAX])
{
#pragma unroll
  for (int s = 0; s < KMAX; ++s) best[s] = kKeyEmpty;
#pragma unroll
  for (int s = 0; s < KMAX; ++s)
    if (s < k)
    {
      knn_key m = L.v[0];
      group_min<G>(m);
      best[s] = m;
      if (L.v[0] == m)  // the owner retires it (keys of real candidates are unique; empty heads all look alike, harmless)
      {
#pragma unroll
        for (int j = 0; j + 1 < KMAX; ++j) L.v[j] = L.v[j + 1];
        L.v[KMAX - 1] = kKeyEmpty;
      }
    }
}

// a query's place in one level of the grid
struct LevelView
{
  int cx, cy, cz;
  float outd2;  // squared distance from the query to the grid box (0 inside), shrunk by a guard factor
};
__device__ __forceinline__ void level_view(LevelView& v, const GridDesc& g, float qx, float qy, float qz)
{
  v.cx = cell_coord(qx, g.origin[0], g.inv_cell, g.dims[0]);
  v.cy = cell_coord(qy, g.origin[1], g.inv_cell, g.dims[1]);
  v.cz = cell_coord(qz, g.origin[2], g.inv_cell, g.dims[2]);
  const float q[3] = {qx, qy, qz};
  float o = 0.f;
#pragma unroll
  for (int d = 0; d < 3; ++d)
  {
    const float lo = g.origin[d], hi = g.origin[d] + g.dims[d] * g.cell;
    float e = 0.f;
    if (q[d] < lo) e = lo - q[d];
    else if (q[d] > hi) e = q[d] - hi;
    o += e * e;
  }
  v.outd2 = o * 0.999f;
}
// every point closer than r cells (minus a 0.1 % guard for the float cell assignment) to the query lies in the block
// of (2 r + 1)^3 cells around the query's cell: the squared radius the block proves
__device__ __forceinline__ float proven2(const GridDesc& g, const LevelView& v, int r)
{
  const float br = ((float)r - 0.001f) * g.cell;
  return v.outd2 + br * br;
}

// The rows of the block of (2 r + 1)^3 cells around a query, dealt to the G lanes of its group (row ri -> lane
// ri % G): bounds of this lane's rows, as offsets into the level's cell-sorted array.  live == false: no rows.
template <int G, int E>
__device__ __forceinline__ void shell_rows(const GridDesc& g, const uint32_t* __restrict__ cs, const LevelView& v, int r, int gl, bool live,
                                           uint32_t (&b)[E], uint32_t (&en)[E], bool& covered)
{
  const int nx = g.dims[0], ny = g.dims[1], nz = g.dims[2];
  const int z0 = max(0, v.cz - r), z1 = min(nz - 1, v.cz + r);
  const int y0 = max(0, v.cy - r), y1 = min(ny - 1, v.cy + r);
  const int x0 = max(0, v.cx - r), x1 = min(nx - 1, v.cx + r);
  const int ys = y1 - y0 + 1;
  const int nrows = live ? (z1 - z0 + 1) * ys : 0;
  covered = (x0 == 0 && y0 == 0 && z0 == 0 && x1 == nx - 1 && y1 == ny - 1 && z1 == nz - 1);
  const int inv_ys = (1 << 16) / ys + 1;  // ri / ys == (ri * inv_ys) >> 16 for ys <= 9, ri < 128: no integer division
#pragma unroll
  for (int e = 0; e < E; ++e)
  {
    const int ri = gl + e * G;
    b[e] = 0; en[e] = 0;
    if (ri < nrows)
    {
      const int zi = (ri * inv_ys) >> 16;
      const int row = ((z0 + zi) * ny + (y0 + ri - zi * ys)) * nx;
      b[e] = cs[row + x0];
      en[e] = cs[row + x1 + 1];
    }
  }
}

// The rows a group fetched, flattened lane by lane (entry f = lane * E + e): start and end of the run in the
// cell-sorted array and the number of candidates in front of it.  Kept in LDS at [slot0 + e][thread].  Returns the
// block's population (uniform across the group).  live == false: nothing is stored, the lane counts as empty.
template <int G, int E>
__device__ __forceinline__ unsigned store_rows(uint32_t* __restrict__ lb, uint32_t* __restrict__ le, uint32_t* __restrict__ lp, int slot0, int tid, int gl,
                                               bool live, const uint32_t (&b)[E], const uint32_t (&en)[E])
{
  unsigned mine = 0;
#pragma unroll
  for (int e = 0; e < E; ++e) mine += en[e] - b[e];
  if (!live) mine = 0;
  unsigned inc = mine;
#pragma unroll
  for (int o = 1; o < G; o <<= 1)
  {
    const unsigned t = (unsigned)__shfl_up((int)inc, o, G);
    if (gl >= o) inc += t;
  }
  const unsigned total = (unsigned)__shfl((int)inc, G - 1, G);
  if (live)
  {
    unsigned run = inc - mine;
#pragma unroll
    for (int e = 0; e < E; ++e)
    {
      lb[(slot0 + e) * 256 + tid] = b[e];
      le[(slot0 + e) * 256 + tid] = en[e];
      lp[(slot0 + e) * 256 + tid] = run;
      run += en[e] - b[e];
    }
  }
  return total;
}

// The scan of a block: its `total` candidates are dealt EVENLY to the G lanes of the group, whatever the rows'
// lengths (lane gl takes the flattened range [gl total / G, (gl + 1) total / G)), every lane walks its range four
// candidates at a time -- the loads of the next four in flight while the current four are compared -- and keeps
// its k best in a sorted list in registers.  E: entries per lane of the block's row table, slot0: where it starts.
template <int KMAX, int G>
__device__ __forceinline__ void scan_rows(LaneList<KMAX>& L, const uint32_t* __restrict__ lb, const uint32_t* __restrict__ le,
                                          const uint32_t* __restrict__ lp, int tid, int gl, int slot0, int E, unsigned total,
                                          const float4* __restrict__ sorted, float qx, float qy, float qz, int* route)
{
  constexpr int U = 4;
  const int gbase = tid - gl;
  const unsigned S = (unsigned)(((unsigned long long)gl * total) / G);
  unsigned n = (unsigned)(((unsigned long long)(gl + 1) * total) / G) - S;  // candidates of this lane
  // entry f of the flattened table lives at [slot0 + f % E][gbase + f / E]; E is one of three constants: the division
  // is a multiplication (exact for f < 5000)
  constexpr int E1 = (9 + G - 1) / G, E2 = (25 + G - 1) / G, E3 = (49 + G - 1) / G;
  const int inv = E == E1 ? (65536 / E1 + 1) : (E == E2 ? (65536 / E2 + 1) : (65536 / E3 + 1));
  auto at = [&](int f) {
    const int lane = (f * inv) >> 16;
    return (slot0 + f - lane * E) * 256 + gbase + lane;
  };
  int f = 0;
  uint32_t c = 0, cend = 0;
  if (n > 0)
  {
    // the last entry whose prefix is <= S (entries in front of it may be empty and share its prefix)
    int lo = 0, hi = G * E - 1;
    while (lo < hi)
    {
      const int mid = (lo + hi + 1) >> 1;
      if (lp[at(mid)] <= S) lo = mid; else hi = mid - 1;
    }
    f = lo;
    const int a0 = at(f);
    c = lb[a0] + (S - lp[a0]);
    cend = le[a0];
  }
  // the next U candidates of the range: addresses (0 = none)
  auto gen = [&](uint32_t (&addr)[U], bool (&ok)[U]) {
#pragma unroll
    for (int u = 0; u < U; ++u)
    {
      ok[u] = n > 0;
      addr[u] = 0;
      if (ok[u])
      {
        while (c >= cend)
        {
          ++f;
          const int a1 = at(f);
          c = lb[a1];
          cend = le[a1];
        }
        addr[u] = c++;
        --n;
      }
    }
  };
  uint32_t addr[U];
  bool ok[U];
  float4 p[U];
  gen(addr, ok);
#pragma unroll
  for (int u = 0; u < U; ++u) p[u] = sorted[addr[u]];
  int walked = 0;
  while (__any(ok[0]))
  {
    float4 cur[U];
    bool has[U];
#pragma unroll
    for (int u = 0; u < U; ++u) { cur[u] = p[u]; has[u] = ok[u]; }
    gen(addr, ok);
    if (__any(ok[0]))
    {
#pragma unroll
      for (int u = 0; u < U; ++u) p[u] = sorted[addr[u]];
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
      if (has[u])
      {
        ++walked;
        const float dx = qx - cur[u].x, dy = qy - cur[u].y, dz = qz - cur[u].z;
        L.insert(make_key((dx * dx + dy * dy) + dz * dz, __float_as_int(cur[u].w)));
      }
  }
  if (route)
  {
    atomicAdd(&route[2], walked);
    atomicMax(&route[5], walked);
  }
}

enum { kOutNone = 0, kOutFound = 1, kOutFar = 2, kOutTail = 3 };

// The exact k nearest neighbours of this group's query: best[0 .. k) ascending by (distance, index), uniform across
// the group.  Every lane of the wavefront calls it; groups without a query (active == false) come back with kOutNone.
template <int KMAX, int G>
__device__ __forceinline__ int group_search(const GridDesc* __restrict__ desc, const GridPtrs& gp, float qx, float qy, float qz, int k, float far_d2,
                                            bool active, int gl, int tid, uint32_t* __restrict__ lb, uint32_t* __restrict__ le, uint32_t* __restrict__ lp,
                                            knn_key (&best)[KMAX], float& ub_out, int* route)
{
  constexpr int E1 = (9 + G - 1) / G, E2 = (25 + G - 1) / G, E3 = (49 + G - 1) / G;
  static_assert(2 * E1 + E2 <= kSlots && E3 <= kSlots, "row slots");
  const GridDesc g0 = desc[0], g1 = desc[1], g2 = desc[2];
  LevelView v0, v1, v2;
  level_view(v0, g0, qx, qy, qz);
  level_view(v1, g1, qx, qy, qz);
  level_view(v2, g2, qx, qy, qz);
  auto bound2 = [&](int s) {
    const int l = shell_level(s), r = shell_r(s);
    return l == 0 ? proven2(g0, v0, r) : (l == 1 ? proven2(g1, v1, r) : proven2(g2, v2, r));
  };
  // rows of the block of shell s (any shell, per group) into slots [0, E3); groups that are not concerned keep what
  // their slots hold (rows of shells 0..2 they may still scan)
  auto fetch_any = [&](int s, bool live, bool& covered) -> unsigned {
    const int l = shell_level(s), r = shell_r(s);
    const GridDesc& g = l == 0 ? g0 : (l == 1 ? g1 : g2);
    const LevelView& v = l == 0 ? v0 : (l == 1 ? v1 : v2);
    const uint32_t* cs = l == 0 ? gp.cell_start[0] : (l == 1 ? gp.cell_start[1] : gp.cell_start[2]);
    uint32_t b[E3], en[E3];
    shell_rows<G, E3>(g, cs, v, r, gl, live, b, en, covered);
    return store_rows<G, E3>(lb, le, lp, 0, tid, gl, live, b, en);
  };

  int outcome = kOutNone;
  int sh = -1, slot0 = 0, ents = E1;
  unsigned total = 0;
  bool sh_covered = false;
  bool need = active;
  ub_out = INFINITY;

  // shells 0, 1, 2: the bounds of all their rows in one memory round trip
  unsigned tot3[3];
  {
    uint32_t b0[E1], e0[E1], b1[E2], e1[E2], b2[E1], e2[E1];
    bool c0, c1, c2;
    shell_rows<G, E1>(g0, gp.cell_start[0], v0, 1, gl, active, b0, e0, c0);
    shell_rows<G, E2>(g0, gp.cell_start[0], v0, 2, gl, active, b1, e1, c1);
    shell_rows<G, E1>(g1, gp.cell_start[1], v1, 1, gl, active, b2, e2, c2);
    tot3[0] = store_rows<G, E1>(lb, le, lp, 0, tid, gl, active, b0, e0);
    tot3[1] = store_rows<G, E2>(lb, le, lp, E1, tid, gl, active, b1, e1);
    tot3[2] = store_rows<G, E1>(lb, le, lp, E1 + E2, tid, gl, active, b2, e2);
    const bool cov[3] = {c0, c1, c2};
    const int first[3] = {0, E1, E1 + E2}, count[3] = {E1, E2, E1};
#pragma unroll
    for (int s = 0; s < 3; ++s)
      if (need)
      {
        if (tot3[s] >= (unsigned)k || cov[s])
        {
          sh = s; sh_covered = cov[s]; slot0 = first[s]; ents = count[s]; total = tot3[s];
          need = false;
        }
        else if (bound2(s) > far_d2)
        {
          // fewer than k points inside a radius beyond the rejection distance
          outcome = kOutFar;
          need = false;
        }
      }
  }
  // sparser neighbourhoods: one larger block after the other, by the counts alone
  for (int s = 3; s < kShells; ++s)
  {
    if (!__any(need)) break;
    bool cov;
    const unsigned tot = fetch_any(s, need, cov);
    if (need)
    {
      if (tot >= (unsigned)k || cov)
      {
        sh = s; sh_covered = cov; slot0 = 0; ents = E3; total = tot;
        need = false;
      }
      else if (bound2(s) > far_d2)
      {
        outcome = kOutFar;
        need = false;
      }
    }
  }
  if (need) outcome = kOutTail;  // not even the largest block holds k points
  if (route && gl == 0 && active)
  {
    if (sh >= 3) atomicAdd(&route[1], 1);
    if (sh == 0) atomicAdd(&route[4], 1);
    if (outcome == kOutFar) atomicAdd(&route[3], 1);
  }

  // first scan
  int sh2 = -1;
  if (__any(sh >= 0))
  {
    LaneList<KMAX> L;
    L.reset();
    const float4* sorted = sh < 2 ? gp.sorted[0] : (sh < 4 ? gp.sorted[1] : gp.sorted[2]);
    scan_rows<KMAX, G>(L, lb, le, lp, tid, gl, slot0, ents, sh >= 0 ? total : 0u, sorted, qx, qy, qz, route);
    merge_lists<KMAX, G>(L, k, best);
    if (sh >= 0)
    {
      const float b2 = bound2(sh);
      int below = 0;
#pragma unroll
      for (int s = 0; s < KMAX; ++s)
        if (s < k && key_d2(best[s]) < b2) ++below;
      if (sh_covered || below >= k) outcome = kOutFound;
      else if (b2 > far_d2) outcome = kOutFar;
      else
      {
        // the k-th best seen bounds the k-th distance: the first block that proves more than it settles the query
        float ub = INFINITY;
#pragma unroll
        for (int s = 0; s < KMAX; ++s)
          if (s == k - 1) ub = key_d2(best[s]);
        ub_out = ub;
        for (int s = sh + 1; s < kShells && sh2 < 0; ++s)
          if (bound2(s) > ub) sh2 = s;
        if (sh2 < 0) outcome = kOutTail;
        else if (route && gl == 0) atomicAdd(&route[0], 1);
      }
    }
  }
  // second scan, certain
  if (__any(sh2 >= 0))
  {
    int slot2 = 0, ents2 = E1;
    unsigned total2 = 0;
    if (sh2 >= 0 && sh2 < 3)
    {
      // still in its slots from the first round trip (this group fetched nothing since)
      slot2 = sh2 == 1 ? E1 : E1 + E2;
      ents2 = sh2 == 1 ? E2 : E1;
      total2 = sh2 == 1 ? tot3[1] : tot3[2];
    }
    if (__any(sh2 >= 3))
    {
      bool cov;
      const unsigned tot = fetch_any(sh2 >= 3 ? sh2 : 3, sh2 >= 3, cov);
      if (sh2 >= 3) { slot2 = 0; ents2 = E3; total2 = tot; }
    }
    LaneList<KMAX> L;
    L.reset();
    const float4* sorted = sh2 < 2 ? gp.sorted[0] : (sh2 < 4 ? gp.sorted[1] : gp.sorted[2]);
    knn_key second[KMAX];
    scan_rows<KMAX, G>(L, lb, le, lp, tid, gl, slot2, ents2, sh2 >= 0 ? total2 : 0u, sorted, qx, qy, qz, route);
    merge_lists<KMAX, G>(L, k, second);
    if (sh2 >= 0)
    {
#pragma unroll
      for (int s = 0; s < KMAX; ++s) best[s] = second[s];
      outcome = kOutFound;
    }
  }
  return outcome;
}

struct FusedType
{
  const float4* queries;  // AoS keypoints, two float4 each
  int nq;
  int block0, nblocks;    // this type's logical blocks [block0, block0 + nblocks)
  int k;
  float far_d2;
  MatchConst mc;
  const GridDesc* desc;
  GridPtrs gp;
  const float4* xyzl;
  int npoints;
  double* rec;
  uint8_t* status;
  int cap;
  int* hist;              // [8] rejection histogram, [8] queries handed to the tail kernel
  int* list;
  float4* list_pts;
  int route_stats;        // diagnostics: count the routes the searches take (hist[10..15])
};
struct FusedArgs
{
  Rigid pose;
  FusedType t[3];
  int nblocks;  // logical blocks of all types
};

constexpr int kNbCols = 64;  // keypoints a block fits models for at most (G >= 4)
struct FusedShared
{
  union
  {
    struct { uint32_t lb[kSlots * 256], le[kSlots * 256], lp[kSlots * 256]; } rows;  // during the search
    struct { float4 nb[kKnnMax * kNbCols]; float nd[kKnnMax * kNbCols]; } edge;  // during the model fit (edges)
  } u;
  int idx[kKnnMax * kNbCols];
  float d2[kKnnMax * kNbCols];
  int cnt[kNbCols];
  int lh[LSA_MATCH_NSTATUS];
  int route[6];  // diagnostics: [0] second scans, [1] first block beyond shell 2, [2] candidates scanned / 64, [3] far, [4] first block = shell 0, [5] longest lane walk
};

template <int KMAX, int G, int TYPE>
__device__ __forceinline__ void fused_type(const Rigid& pose, const FusedType& t, int block, FusedShared& sh)
{
  constexpr int QB = 256 / G;
  static_assert(QB <= kNbCols, "LDS columns");
  const int tid = threadIdx.x, gl = tid % G, ql = tid / G;
  const int q = block * QB + ql;
  const bool active = q < t.nq && !t.mc.bad_param;
  if (tid < LSA_MATCH_NSTATUS) sh.lh[tid] = 0;
  if (tid < 6) sh.route[tid] = 0;
  if (t.route_stats) __syncthreads();
  float qx = 0.f, qy = 0.f, qz = 0.f;
  if (active)
  {
    // KeypointsMatcher: worldPoint = PosePrior * basePoint in double, narrowed to float for the search
    // (KeypointsMatcher.cxx:117-118, KDTreePCLAdaptor.h:96-100)
    const float4 q4 = t.queries[2 * (size_t)q];
    double wx, wy, wz;
    rigid_apply(pose, (double)q4.x, (double)q4.y, (double)q4.z, wx, wy, wz);
    qx = (float)wx; qy = (float)wy; qz = (float)wz;
  }
  knn_key best[KMAX];
#pragma unroll
  for (int s = 0; s < KMAX; ++s) best[s] = kKeyEmpty;
  float ub = INFINITY;
  const int outcome = group_search<KMAX, G>(t.desc, t.gp, qx, qy, qz, t.k, t.far_d2, active, gl, tid, sh.u.rows.lb, sh.u.rows.le, sh.u.rows.lp, best, ub, t.route_stats ? sh.route : nullptr);
  if (gl == 0)
  {
    int cnt = 0;
#pragma unroll
    for (int s = 0; s < KMAX; ++s)
      if (s < t.k)
      {
        sh.idx[s * kNbCols + ql] = key_idx(best[s]);
        sh.d2[s * kNbCols + ql] = key_d2(best[s]);
        if (best[s] != kKeyEmpty) ++cnt;
      }
    if (outcome == kOutTail)
    {
      // handed to the tail kernel: the query in target coordinates and what is known about its k-th distance
      const int slot = atomicAdd(t.hist + LSA_MATCH_NSTATUS, 1);
      t.list[slot] = q;
      t.list_pts[slot] = make_float4(qx, qy, qz, ub);
      cnt = kPending;
    }
    else if (outcome == kOutFar) cnt = kKnnFar;
    sh.cnt[ql] = cnt;
  }
  __syncthreads();  // the search is over for the whole block: its row slots become the edge staging area
  if (tid < QB)
  {
    const int i = block * QB + tid;
    if (i < t.nq)
    {
      const int n = t.mc.bad_param ? 0 : sh.cnt[tid];
      if (n != kPending)
      {
        const int st = fit_model<KMAX, TYPE>(
          t.queries[2 * (size_t)i], t.mc, n, [&](int s) { return sh.idx[s * kNbCols + tid]; }, [&](int s) { return sh.d2[s * kNbCols + tid]; }, t.xyzl,
          sh.u.edge.nb, sh.u.edge.nd, kNbCols, tid, t.rec, t.cap, i);
        t.status[i] = (uint8_t)st;
        atomicAdd(&sh.lh[st], 1);
      }
    }
  }
  __syncthreads();
  if (tid < LSA_MATCH_NSTATUS && sh.lh[tid]) atomicAdd(&t.hist[tid], sh.lh[tid]);
  if (t.route_stats && tid < 6 && sh.route[tid])
  {
    if (tid == 5) atomicMax(&t.hist[LSA_MATCH_NSTATUS + 2 + tid], sh.route[tid]);
    else atomicAdd(&t.hist[LSA_MATCH_NSTATUS + 2 + tid], sh.route[tid]);
  }
}

// lanes per query of the three types
constexpr int kGE = 8, kGB = 8;

template <int KE, int KP, int KB, int GP>
__global__ __launch_bounds__(256) void k_match_fused(FusedArgs a)
{
  __shared__ FusedShared sh;
  // consecutive logical blocks hold neighbouring keypoints (scan order): they are dealt to ONE XCD, whose L2 then
  // serves a compact region of the target (hardware blocks b, b + 8, ... share an XCD)
  const int per_xcd = (a.nblocks + 7) / 8;
  const int block = (blockIdx.x % 8) * per_xcd + blockIdx.x / 8;
  if (block >= a.nblocks) return;
  if (block < a.t[1].block0) fused_type<KE, kGE, LSA_EDGE>(a.pose, a.t[0], block - a.t[0].block0, sh);
  else if (block < a.t[2].block0) fused_type<KP, GP, LSA_PLANE>(a.pose, a.t[1], block - a.t[1].block0, sh);
  else fused_type<KB, kGB, LSA_BLOB>(a.pose, a.t[2], block - a.t[2].block0, sh);
}

// Tail: the queries no block of the grid could settle, one wavefront each over the whole target (every lane
// keeps the k best of its share, one merge at the end), then their model fit.
constexpr int kTailBlocks = 16;  // per type
template <int KMAX, int TYPE>
__device__ __forceinline__ void tail_type(const FusedType& t, int block, FusedShared& sh)
{
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  if (tid < LSA_MATCH_NSTATUS) sh.lh[tid] = 0;
  __syncthreads();
  const int nwork = t.nq > 0 ? t.hist[LSA_MATCH_NSTATUS] : 0;
  const float4* __restrict__ sorted = t.gp.sorted[0];
  for (int w = block * 4 + wv; w < nwork; w += kTailBlocks * 4)
  {
    const int q = t.list[w];
    const float4 qp = t.list_pts[w];
    LaneList<KMAX> L;
    L.reset();
    for (int c = lane; c < t.npoints; c += 64)
    {
      const float4 p = sorted[c];
      const float dx = qp.x - p.x, dy = qp.y - p.y, dz = qp.z - p.z;
      L.insert(make_key((dx * dx + dy * dy) + dz * dz, __float_as_int(p.w)));
    }
    knn_key best[KMAX];
    merge_lists<KMAX, 64>(L, t.k, best);
    if (lane == 0)
    {
      int cnt = 0;
#pragma unroll
      for (int s = 0; s < KMAX; ++s)
        if (s < t.k)
        {
          sh.idx[s * kNbCols + wv] = key_idx(best[s]);
          sh.d2[s * kNbCols + wv] = key_d2(best[s]);
          if (best[s] != kKeyEmpty) ++cnt;
        }
      atomicAdd(t.hist + LSA_MATCH_NSTATUS + 1, 1);
      const int st = fit_model<KMAX, TYPE>(
        t.queries[2 * (size_t)q], t.mc, cnt, [&](int s) { return sh.idx[s * kNbCols + wv]; }, [&](int s) { return sh.d2[s * kNbCols + wv]; }, t.xyzl,
        sh.u.edge.nb, sh.u.edge.nd, kNbCols, wv, t.rec, t.cap, q);
      t.status[q] = (uint8_t)st;
      atomicAdd(&sh.lh[st], 1);
    }
  }
  __syncthreads();
  if (tid < LSA_MATCH_NSTATUS && sh.lh[tid]) atomicAdd(&t.hist[tid], sh.lh[tid]);
}

template <int KE, int KP, int KB>
__global__ __launch_bounds__(256) void k_match_tail(FusedArgs a)
{
  __shared__ FusedShared sh;
  const int type = blockIdx.x / kTailBlocks, block = blockIdx.x % kTailBlocks;
  if (type == 0) tail_type<KE, LSA_EDGE>(a.t[0], block, sh);
  else if (type == 1) tail_type<KP, LSA_PLANE>(a.t[1], block, sh);
  else tail_type<KB, LSA_BLOB>(a.t[2], block, sh);
}

template <int KE, int KP, int KB>
void launch_fused(lsa_ctx* ctx, const FusedArgs& a, int plane_lanes, double bytes, hipStream_t st)
{
  const int grid = 8 * ((a.nblocks + 7) / 8);
  {
    ProfScope ps(ctx, "match_fused", bytes, st);
    if (plane_lanes >= 8) hipLaunchKernelGGL((k_match_fused<KE, KP, KB, 8>), dim3(grid), dim3(256), 0, st, a);
    else hipLaunchKernelGGL((k_match_fused<KE, KP, KB, 4>), dim3(grid), dim3(256), 0, st, a);
  }
  ProfScope ps(ctx, "match_tail", 0., st);
  hipLaunchKernelGGL((k_match_tail<KE, KP, KB>), dim3(3 * kTailBlocks), dim3(256), 0, st, a);
}

}  // namespace

namespace lsa
{

// Enqueues the matches `preps` describes (at most one per keypoint type, every one with a non-empty target and at
// least one keypoint) as one launch + the tail launch on `st`.
int enqueue_fused_match(lsa_ctx* ctx, const MatchPrep* preps, int count, const double pose[16], hipStream_t st)
{
  FusedArgs a;
  std::memset(&a, 0, sizeof(a));
  row_major_to_rt(pose, a.pose.R, a.pose.t);
  int kmax[3] = {1, 1, 1};
  double bytes = 0;
  for (int i = 0; i < count; ++i)
  {
    const MatchPrep& p = preps[i];
    FusedType& t = a.t[p.type];
    Target& tg = ctx->target[p.ti];
    MatchBuf& mb = ctx->match[p.type];
    t.queries = reinterpret_cast<const float4*>(p.queries);
    t.nq = p.nq;
    t.k = p.mc.k;
    t.far_d2 = p.far_d2;
    t.mc = p.mc;
    t.desc = tg.desc;
    for (int l = 0; l < kGridLevels; ++l) { t.gp.cell_start[l] = tg.lv[l].cell_start; t.gp.sorted[l] = tg.lv[l].sorted; }
    t.xyzl = tg.xyzl;
    t.npoints = tg.m;
    t.rec = mb.rec; t.status = mb.status; t.cap = mb.cap;
    t.hist = p.hist;
    t.list = mb.slow_list; t.list_pts = mb.slow_pts;
    t.route_stats = ctx->route_stats ? 1 : 0;
    kmax[p.type] = p.mc.k;
    // algorithmic bytes (B_icp): keypoint in, k gathered target points, residual record out
    bytes += (double)p.nq * (32 + p.mc.k * 32 + 136);
  }
  const int lanes[3] = {kGE, ctx->knn_lanes[LSA_PLANE] >= 8 ? 8 : 4, kGB};
  int block0 = 0;
  for (int k = 0; k < 3; ++k)
  {
    a.t[k].block0 = block0;
    a.t[k].nblocks = a.t[k].nq > 0 ? (int)(((size_t)a.t[k].nq * lanes[k] + 255) / 256) : 0;
    block0 += a.t[k].nblocks;
  }
  a.nblocks = block0;
  if (block0 == 0) return LSA_OK;
  const int ke = kmax[0], kp = kmax[1], kb = kmax[2];
  if (ke <= 8 && kp <= 5) launch_fused<8, 5, 16>(ctx, a, lanes[1], bytes, st);
  else if (kp <= 5) launch_fused<16, 5, 16>(ctx, a, lanes[1], bytes, st);
  else if (kp <= 8) launch_fused<16, 8, 16>(ctx, a, lanes[1], bytes, st);
  else launch_fused<16, 16, 16>(ctx, a, lanes[1], bytes, st);
  (void)kb;
  return LSA_OK;
}

}  // namespace lsa
