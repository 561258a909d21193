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

constexpr int kPending = -2;  // neighbour count of a query no block of the grid settles
constexpr int kShells = 7;
#ifndef LSA_KFILL
#define LSA_KFILL 3
#endif
#ifndef LSA_KHEAVY
#define LSA_KHEAVY 192
#endif
constexpr int kFill = LSA_KFILL;       // a block is scanned first when it holds kFill x k points
constexpr int kHeavyCandidates = LSA_KHEAVY;  // a block with more candidates than this is scanned by the whole wavefront

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
#ifdef LSA_ABLATE_INSERT
    if (key < v[0]) v[0] = key;  // (timing experiment only: wrong results)
    return;
#endif
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
__device__ __forceinline__ void merge_lists(LaneList<KMAX>& L, int k, knn_key (&best)[KMAX], bool take = true)
{
  // take == false: the group is not concerned (it takes part in the exchanges with empty lists) and keeps its best
#pragma unroll
  for (int s = 0; s < KMAX; ++s)
    if (take) best[s] = kKeyEmpty;
#pragma unroll
  for (int s = 0; s < KMAX; ++s)
    if (s < k)
    {
      knn_key m = L.v[0];
      group_min<G>(m);
      if (take) best[s] = m;
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

// inclusive scan / maximum over the G lanes of a group (G <= 16: DPP inside a row of 16 lanes, no LDS round trip)
template <int G>
__device__ __forceinline__ unsigned group_scan(unsigned v, int gl)
{
  static_assert(G <= 16, "groups live inside a DPP row");
  if (G > 1) { const unsigned t = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, true); if (gl >= 1) v += t; }  // row_shr:1
  if (G > 2) { const unsigned t = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xF, 0xF, true); if (gl >= 2) v += t; }
  if (G > 4) { const unsigned t = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xF, true); if (gl >= 4) v += t; }
  if (G > 8) { const unsigned t = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xF, 0xF, true); if (gl >= 8) v += t; }
  return v;
}
template <int G>
__device__ __forceinline__ unsigned group_max(unsigned v)
{
  auto step = [&](auto ctrl) {
    const unsigned o = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, decltype(ctrl)::value, 0xF, 0xF, true);
    v = o > v ? o : v;
  };
  if (G >= 2) step(std::integral_constant<int, 0xB1>());
  if (G >= 4) step(std::integral_constant<int, 0x4E>());
  if (G >= 8) step(std::integral_constant<int, 0x141>());
  if (G >= 16) step(std::integral_constant<int, 0x140>());
  return v;
}

// A group's table of candidate runs in LDS: the NON-EMPTY rows of a block, one entry each -- start and end of the run in
// the cell-sorted array and the number of candidates in front of it -- followed by one (0, 0, total) sentinel.
struct RowTable
{
  uint32_t* b;
  uint32_t* e;
  uint32_t* p;
};
template <int G>
__device__ __forceinline__ constexpr int table_capacity()
{
  constexpr int E1 = (9 + G - 1) / G, E2 = (25 + G - 1) / G, E3 = (49 + G - 1) / G;
  return (G * (2 * E1 + E2) + 3) > (G * E3 + 1) ? (G * (2 * E1 + E2) + 3) : (G * E3 + 1);
}
constexpr int kTableEntries = 2144;  // a block's tables: 256 / G groups x table_capacity<G>() (G = 8: 32 x 67, G = 16: 16 x 67)

// Appends this lane's non-empty rows to the group's table at `base` (lane after lane, so that the table is
// contiguous).  Returns (entries << 20 | candidates) of the whole block, uniform across the group.  live == false:
// the lane contributes nothing and writes nothing.
template <int G, int E>
__device__ __forceinline__ unsigned store_rows(const RowTable& tb, int base, int gl, bool live, const uint32_t (&b)[E], const uint32_t (&en)[E])
{
  unsigned mine = 0;
#pragma unroll
  for (int e = 0; e < E; ++e)
    if (en[e] > b[e]) mine += (1u << 20) + (en[e] - b[e]);
  if (!live) mine = 0;
  const unsigned inc = group_scan<G>(mine, gl);
  const unsigned total = group_max<G>(inc);  // the scan does not decrease along the group: its last value
  if (live)
  {
    unsigned run = inc - mine;
#pragma unroll
    for (int e = 0; e < E; ++e)
      if (en[e] > b[e])
      {
        const int pos = base + (int)(run >> 20);
        tb.b[pos] = b[e];
        tb.e[pos] = en[e];
        tb.p[pos] = run & 0xfffffu;
        run += (1u << 20) + (en[e] - b[e]);
      }
    if (gl == G - 1)
    {
      const int pos = base + (int)(total >> 20);
      tb.b[pos] = 0;
      tb.e[pos] = 0;
      tb.p[pos] = total & 0xfffffu;
    }
  }
  return total;
}

// The scan of a block: its `total` candidates, in the order of the table, are dealt to the G lanes of the group ROUND
// ROBIN (candidate c -> lane c mod G): the lanes of a group read neighbouring addresses of the cell-sorted array -- one
// or two cache lines per load instruction and group where contiguous per-lane runs touched G (the kernel is bound by
// the gathers' cache-line throughput in the vector memory pipeline, not by arithmetic) -- and the k nearest spread over
// the lanes.  Every lane walks U candidates per turn, the loads of the next turn in flight while the current one is
// compared, and follows the table's runs by itself (one entry = start, end, candidates in front; all rows non-empty, a
// (0, 0, total) sentinel behind them).  It keeps its k best in a sorted list in registers.  base: the group's table of
// the block, nent: its entries.
template <int KMAX, int G>
__device__ __forceinline__ void scan_rows(LaneList<KMAX>& L, const RowTable& tb, int base, int nent, unsigned total, int gl,
                                          const float4* __restrict__ sorted, float qx, float qy, float qz, int* route)
{
  constexpr int U = KMAX >= 8 ? 2 : 4;  // edge lists (k = 8, 10, 16): fewer candidates in flight, fewer registers (with 4 the k = 8 kernel spilled 57)
  int f = base;                 // next entry of the table
  uint32_t pend = 0, roff = 0;  // the current run: candidates [.., pend) of the block, candidate c at address roff + c
  uint32_t cnext = (uint32_t)gl;
  // the next U candidates of this lane: addresses (0 = none)
  auto gen = [&](uint32_t (&addr)[U], bool (&ok)[U]) {
#pragma unroll
    for (int u = 0; u < U; ++u)
    {
      const uint32_t c = cnext + (uint32_t)(u * G);
      ok[u] = c < total;
      addr[u] = 0;
      if (ok[u])
      {
        while (c >= pend)
        {
          const uint32_t rb = tb.b[f], re = tb.e[f], rp = tb.p[f];
          roff = rb - rp;
          pend = rp + (re - rb);
          ++f;
        }
        addr[u] = roff + c;
      }
    }
    cnext += (uint32_t)(U * G);
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

// A block of the coarse levels next to a dense part of the target can hold a thousand candidates, and the launch lasts
// as long as its longest walk.  Such a block is scanned by the WHOLE wavefront, one heavy group's after the other: the
// group's table is in LDS, its query travels by lane reads, every lane keeps its k best and the 64 lists merge into the
// group's slot in LDS -- where the lists all groups hold so far wait meanwhile (the caller puts them there and takes
// them back: the registers go to the wavefront's lists).  Inlined: as a function of its own it saved and restored fifty
// registers through scratch memory on every call -- 5 of the 11 MB a launch wrote (PMC WRITE_SIZE, scripts/pmc_microbench.sh).
template <int KMAX>
__device__ __forceinline__ void scan_heavy_blocks(unsigned long long pending, int G, const RowTable tb, knn_key* save, const float4* s0, const float4* s1,
                                               const float4* s2, int cur, int cbase, int nent, unsigned cand, float qx, float qy, float qz, int k, int* route)
{
  const int tid = threadIdx.x, lane = tid & 63;
  while (pending)
  {
    const int lead = __ffsll((long long)pending) - 1;
    pending &= pending - 1;
    const int hcur = __shfl(cur, lead), hbase = __shfl(cbase, lead), hnent = __shfl(nent, lead);
    const unsigned hcand = (unsigned)__shfl((int)cand, lead);
    const float hx = __shfl(qx, lead), hy = __shfl(qy, lead), hz = __shfl(qz, lead);
    const float4* hsorted = hcur < 2 ? s0 : (hcur < 4 ? s1 : s2);
    knn_key* const hslot = save + (size_t)(((tid & ~63) + lead) / G) * KMAX;
    LaneList<KMAX> H;
    H.reset();
    scan_rows<KMAX, 64>(H, tb, hbase, hnent, hcand, lane, hsorted, hx, hy, hz, nullptr);
#pragma unroll
    for (int s = 0; s < KMAX; ++s)
    {
      knn_key m = kKeyEmpty;
      if (s < k)
      {
        m = H.v[0];
        group_min<64>(m);
        if (H.v[0] == m)
        {
#pragma unroll
          for (int j = 0; j + 1 < KMAX; ++j) H.v[j] = H.v[j + 1];
          H.v[KMAX - 1] = kKeyEmpty;
        }
      }
      if (lane == 0) hslot[s] = m;
    }
    if (route && lane == 0) atomicAdd(&route[2], (int)hcand);
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the same wavefront wrote the slots: in order, only the counter to wait for
}

enum { kOutNone = 0, kOutFound = 1, kOutFar = 2, kOutTail = 3 };

// The exact k nearest neighbours of this group's query: best[0 .. k) ascending by (distance, index), uniform across
// the group.  Every lane of the wavefront calls it; groups without a query (active == false) come back with kOutNone.
template <int KMAX, int G>
__device__ __forceinline__ int group_search(const GridDesc* __restrict__ desc, const GridPtrs& gp, float qx, float qy, float qz, int k, float far_d2,
                                            bool active, int gl, int tid, const RowTable& tb, knn_key* save, knn_key (&best)[KMAX], float& ub_out, int* route,
                                            unsigned long long* phase = nullptr)
{
  auto stamp = [&](int i) { if (phase && (tid & 63) == 0) phase[(tid >> 6) * 8 + i] = wall_clock64(); };
  constexpr int E1 = (9 + G - 1) / G, E2 = (25 + G - 1) / G, E3 = (49 + G - 1) / G;
  constexpr int CAP = table_capacity<G>();
  static_assert((256 / G) * CAP <= kTableEntries, "row tables");
  const int gbase = ((tid - gl) / G) * CAP;  // this group's tables
  const int base3[3] = {gbase, gbase + G * E1 + 1, gbase + G * (E1 + E2) + 2};
  const GridDesc g0 = desc[0], g1 = desc[1], g2 = desc[2];
  LevelView v0, v1, v2;
  level_view(v0, g0, qx, qy, qz);
  level_view(v1, g1, qx, qy, qz);
  level_view(v2, g2, qx, qy, qz);
  auto bound2 = [&](int s) {
    const int l = shell_level(s), r = shell_r(s);
    return l == 0 ? proven2(g0, v0, r) : (l == 1 ? proven2(g1, v1, r) : proven2(g2, v2, r));
  };
  // rows of the block of shell s (any shell, per group) into the table at gbase; groups that are not concerned keep
  // what their tables hold (blocks of shells 0..2 they may still scan)
  auto fetch_any = [&](int s, bool live, bool& covered) -> unsigned {
    const int l = shell_level(s), r = shell_r(s);
    const GridDesc& g = l == 0 ? g0 : (l == 1 ? g1 : g2);
    const LevelView& v = l == 0 ? v0 : (l == 1 ? v1 : v2);
    const uint32_t* cs = l == 0 ? gp.cell_start[0] : (l == 1 ? gp.cell_start[1] : gp.cell_start[2]);
    uint32_t b[E3], en[E3];
    shell_rows<G, E3>(g, cs, v, r, gl, live, b, en, covered);
    return store_rows<G, E3>(tb, gbase, gl, live, b, en);
  };

  int outcome = kOutNone;
  int sh = -1, base = gbase;
  unsigned total = 0;  // (entries << 20 | candidates) of the block to scan
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
    tot3[0] = store_rows<G, E1>(tb, base3[0], gl, active, b0, e0);
    tot3[1] = store_rows<G, E2>(tb, base3[1], gl, active, b1, e1);
    tot3[2] = store_rows<G, E1>(tb, base3[2], gl, active, b2, e2);
    const bool cov[3] = {c0, c1, c2};
    // fewer than k points inside a radius beyond the rejection distance
#pragma unroll
    for (int j = 0; j < 3; ++j)
      if (need && !cov[j] && (tot3[j] & 0xfffffu) < (unsigned)k && bound2(j) > far_d2)
      {
        outcome = kOutFar;
        need = false;
      }
    auto choose = [&](int j) {
      sh = j; sh_covered = cov[j]; base = base3[j]; total = tot3[j];
      need = false;
    };
    // A block that holds just k points seldom proves them nearest (they sit in its corners): the first block with a
    // few times k is the one to scan -- most queries then need no second scan; failing that, the largest of the three
    // that holds k at all.
#pragma unroll
    for (int j = 0; j < 3; ++j)
      if (need && (cov[j] || (tot3[j] & 0xfffffu) >= (unsigned)(kFill * k))) choose(j);
#pragma unroll
    for (int j = 2; j >= 0; --j)
      if (need && (tot3[j] & 0xfffffu) >= (unsigned)k) choose(j);
  }
  // sparser neighbourhoods: one larger block after the other, by the counts alone
  for (int s = 3; s < kShells; ++s)
  {
    if (!__any(need)) break;
    bool cov;
    const unsigned tot = fetch_any(s, need, cov);
    if (need)
    {
      if ((tot & 0xfffffu) >= (unsigned)k || cov)
      {
        sh = s; sh_covered = cov; base = gbase; total = tot;
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
  stamp(0);
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
    const unsigned cand = sh >= 0 ? (total & 0xfffffu) : 0u;
    const int nent = sh >= 0 ? (int)(total >> 20) : 0;
    const bool heavy = cand > (unsigned)kHeavyCandidates;
    const unsigned long long pending = __ballot(heavy && gl == 0);
    if (pending)  // nothing held yet: only the heavy groups' results come back
      scan_heavy_blocks<KMAX>(pending, G, tb, save, gp.sorted[0], gp.sorted[1], gp.sorted[2], sh, base, nent, cand, qx, qy, qz, k, route);
    LaneList<KMAX> L;
    L.reset();
    const float4* sorted = sh < 2 ? gp.sorted[0] : (sh < 4 ? gp.sorted[1] : gp.sorted[2]);
    scan_rows<KMAX, G>(L, tb, base, heavy ? 0 : nent, heavy ? 0u : cand, gl, sorted, qx, qy, qz, route);
    stamp(1);
    merge_lists<KMAX, G>(L, k, best, !heavy);
    if (heavy)
    {
      const knn_key* slot = save + (size_t)((tid - gl) / G) * KMAX;
#pragma unroll
      for (int s = 0; s < KMAX; ++s) best[s] = slot[s];
    }
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
  stamp(2);
  // second scan, certain
  if (__any(sh2 >= 0))
  {
    int base2 = gbase;
    unsigned total2 = 0;
    // shells 1 and 2 are still in their tables from the first round trip: a group that scans one of them second has
    // scanned shell 0 or 1 first and fetched nothing since
    const bool kept = sh2 >= 0 && sh2 < 3;
    if (kept)
    {
      base2 = sh2 == 1 ? base3[1] : base3[2];
      total2 = sh2 == 1 ? tot3[1] : tot3[2];
    }
    const bool fetch2 = sh2 >= 0 && !kept;
    if (__any(fetch2))
    {
      bool cov;
      const unsigned tot = fetch_any(fetch2 ? sh2 : 3, fetch2, cov);
      if (fetch2) { base2 = gbase; total2 = tot; }
    }
    const unsigned cand = sh2 >= 0 ? (total2 & 0xfffffu) : 0u;
    const int nent = sh2 >= 0 ? (int)(total2 >> 20) : 0;
    const bool heavy = cand > (unsigned)kHeavyCandidates;
    const unsigned long long pending = __ballot(heavy && gl == 0);
    if (pending)
      scan_heavy_blocks<KMAX>(pending, G, tb, save, gp.sorted[0], gp.sorted[1], gp.sorted[2], sh2, base2, nent, cand, qx, qy, qz, k, route);
    LaneList<KMAX> L;
    L.reset();
    const float4* sorted = sh2 < 2 ? gp.sorted[0] : (sh2 < 4 ? gp.sorted[1] : gp.sorted[2]);
    scan_rows<KMAX, G>(L, tb, base2, heavy ? 0 : nent, heavy ? 0u : cand, gl, sorted, qx, qy, qz, route);
    merge_lists<KMAX, G>(L, k, best, sh2 >= 0 && !heavy);
    if (heavy)  // the second block holds the first: its k best are the answer
    {
      const knn_key* slot = save + (size_t)((tid - gl) / G) * KMAX;
#pragma unroll
      for (int s = 0; s < KMAX; ++s) best[s] = slot[s];
    }
    if (sh2 >= 0) outcome = kOutFound;
  }
  stamp(3);
  return outcome;
}

struct FusedType
{
  const float4* queries;  // AoS keypoints, two float4 each
  int nq;
  int nblocks;            // search blocks of this type
  int mblocks;            // model blocks of this type
  int k;
  float far_d2;
  MatchConst mc;
  const GridDesc* desc;
  GridPtrs gp;
  const float4* xyzl;
  int npoints;
  int* knn_idx;           // [k][cap] neighbour lists: written by the search, read by the model fit
  float* knn_d2;
  int* knn_cnt;
  double* rec;
  uint8_t* status;
  int cap;
  int* hist;              // [8] rejection histogram, [8] diagnostics
  int route_stats;        // diagnostics: count the routes the searches take (hist[10..15])
  unsigned long long* trace;  // diagnostics: per hardware block {start, end, 0, where} (100 MHz clock)
};
struct FusedArgs
{
  Rigid pose;
  FusedType t[3];
  int fuse_model;  // the search kernel fits the models of its own keypoints (no second launch)
  int undistort;   // ... and first moves every keypoint by the motion `ic` interpolated at its own time (lsa_undistort), in place
  InterpConst ic;
  const IcpGate* gate;  // not null: enqueued ahead of its inputs (lsa_icp_gate) -- pose and `ic` come from there, or nothing is done
};

struct SearchShared
{
  uint32_t b[kTableEntries], e[kTableEntries], p[kTableEntries];  // the groups' row tables
  knn_key save[(256 / 8) * 16];  // the groups' lists while the wavefront scans a heavy block for one of them
  int route[6];  // diagnostics: [0] second scans, [1] first block beyond shell 2, [2] candidates walked, [3] far, [4] first block = shell 0, [5] longest lane walk
  unsigned long long phase[32];  // diagnostics (first lane of every wavefront, 8 each): clock after the rows, the first scan, its merge, the second scan | the staging, the barrier, the finish
};

// What the searching lanes leave in LDS for the lane that finishes a keypoint's match (one-launch form): the running
// sums of pcl::computeMeanAndCovarianceMatrix over the neighbours the model keeps, and what was decided on the way.
constexpr int kStagePending = -2;  // no block of the grid settled the query: the finishing wavefront searches the whole target
template <int QB, int KC>
struct FitStage
{
  double acc[9][QB];     // xx xy xz yy yz zz x y z
  float4 q4[QB];         // the keypoint in BASE coordinates (undistorted when the launch undistorts)
  float4 cand[KC][QB];   // the neighbours' points (x, y, z, laser id), ascending (distance, index)
  float last_d2[QB];     // squared distance of the last neighbour kept
  int nsel[QB];          // neighbours kept
  int pre[QB];           // status decided before the PCA (LSA_MATCH_SUCCESS: go on), or kStagePending
  int lh[LSA_MATCH_NSTATUS];
};

// The part of KeypointsMatcher::BuildLineMatch / BuildPlaneMatch / BuildBlobMatch (KeypointsMatcher.cxx:106-346) in front of
// the PCA, by the G lanes that searched the keypoint: the k neighbours' points gathered side by side (one lane each),
// the neighbourhood filter -- GetPerRingLineNeighbors `:349-405` one candidate per lane, GetRansacLineNeighbors
// `:408-480` one line hypothesis per lane -- and the running sums of the covariance, one SUM per lane, every sum over
// the kept neighbours in ascending order as the reference adds them.  best[0 .. n): the neighbours, uniform across the
// group; n: their number (kKnnFar: the k-th lies beyond the rejection distance).  write: this group's lanes store.
template <int KMAX, int TYPE, int G, int QB, int KC>
__device__ __forceinline__ void group_stage(const knn_key (&best)[KMAX], int n, const MatchConst& c, const float4* __restrict__ xyzl, int gl, int slot,
                                            bool live, bool write, FitStage<QB, KC>& fs)
{
  static_assert(KMAX <= KC && KMAX <= 2 * G, "two candidates per lane at most");
  constexpr int E = (KMAX + G - 1) / G;
  const int nn = n > 0 ? n : 0;
  // gather: lane gl fetches neighbours gl, gl + G
#pragma unroll
  for (int e = 0; e < E; ++e)
  {
    knn_key key = kKeyEmpty;
#pragma unroll
    for (int j = 0; j < G; ++j)
      if (e * G + j < KMAX && gl == j) key = best[e * G + j];
    const int s = e * G + gl;
    if (live && write && s < nn && s < KMAX) fs.cand[s][slot] = xyzl[key_idx(key)];
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");  // the group's own lanes wrote them: same wavefront, in order
  unsigned mask = 0;  // neighbours kept, uniform across the group
  int pre = LSA_MATCH_SUCCESS;
  const int gbase = (threadIdx.x & 63) - gl;  // first lane of the group in its wavefront
  auto group_bits = [&](bool v) { return (unsigned)((__ballot(v) >> gbase) & ((1ull << G) - 1ull)); };
  if (TYPE == LSA_EDGE)
  {
    if (c.single_edge_per_ring)
    {
      // GetPerRingLineNeighbors: drop the closest point's own ring and rings more than 4 away, keep the nearest point of
      // every remaining ring (a candidate whose ring showed up earlier is out, whatever became of the earlier one)
      const int closest = nn > 0 ? (int)__float_as_uint(fs.cand[0][slot].w) : 0;
#pragma unroll
      for (int e = 0; e < E; ++e)
      {
        const int t = e * G + gl;
        bool keep = false;
        if (t < nn && t < KMAX)
        {
          const int lid = (int)__float_as_uint(fs.cand[t][slot].w);
          keep = (lid != closest) && (abs(closest - lid) <= 4);
#pragma unroll
          for (int s = 0; s < KMAX - 1; ++s)
            if (s < t && (int)__float_as_uint(fs.cand[s][slot].w) == lid) keep = false;
        }
        mask |= group_bits(keep) << (e * G);
      }
    }
    else if (nn >= 2)
    {
      // GetRansacLineNeighbors: the line through the closest point and candidate pi, for every pi at once
      const float4 p1 = fs.cand[0][slot];
      const Vec3<float> P1 = {p1.x, p1.y, p1.z};
      auto inlier = [&](int ci, const Vec3<float>& dir) {
        const float4 pc = fs.cand[ci][slot];
        return vsqnorm(vcross(vsub(Vec3<float>{pc.x, pc.y, pc.z}, P1), dir)) < c.ransac_sq_inlier;
      };
      unsigned vote = 0;  // (inliers << 8) | (255 - pi): the maximum is the first hypothesis with the most inliers
#pragma unroll
      for (int e = 0; e < E; ++e)
      {
        const int pi = e * G + gl;
        if (pi >= 1 && pi < nn && pi < KMAX)
        {
          const float4 p2 = fs.cand[pi][slot];
          const Vec3<float> dir = normalized3(vsub(Vec3<float>{p2.x, p2.y, p2.z}, P1));
          int cnt = 0;
#pragma unroll
          for (int ci = 1; ci < KMAX; ++ci)
            if (ci < nn && (ci == pi || inlier(ci, dir))) ++cnt;
          const unsigned v = ((unsigned)cnt << 8) | (unsigned)(255 - pi);
          vote = v > vote ? v : vote;
        }
      }
      vote = group_max<G>(vote);
      const int bestpi = 255 - (int)(vote & 0xffu);
      const float4 pb = fs.cand[bestpi][slot];
      const Vec3<float> dir = normalized3(vsub(Vec3<float>{pb.x, pb.y, pb.z}, P1));
#pragma unroll
      for (int e = 0; e < E; ++e)
      {
        const int ci = e * G + gl;
        const bool in = ci < nn && ci < KMAX && (ci == 0 || ci == bestpi || inlier(ci, dir));
        mask |= group_bits(in) << (e * G);
      }
    }
  }
  else
  {
    // n == kKnnFar: the target holds >= k points but the k-th nearest is beyond MaxNeighborsDistance
    if (n == kKnnFar) pre = LSA_MATCH_NEIGHBORS_TOO_FAR;
    else if (n < c.k) pre = LSA_MATCH_NOT_ENOUGH_NEIGHBORS;
    else mask = (1u << c.k) - 1u;
  }
  const int nsel = __popc(mask);
  float last_d2 = 0.f;
#pragma unroll
  for (int s = 0; s < KMAX; ++s)
    if ((mask >> s) & 1u) last_d2 = key_d2(best[s]);
  if (TYPE == LSA_EDGE && nsel < c.min_neighbors) pre = LSA_MATCH_NOT_ENOUGH_NEIGHBORS;
  if (pre == LSA_MATCH_SUCCESS && (double)last_d2 > c.max_dist2) pre = LSA_MATCH_NEIGHBORS_TOO_FAR;
  // the nine sums, lane j < 6 the product j, lanes 6 and 7 the sums of x and y, every lane the sum of z (lane 0's is kept)
  const int j = gl & 7;
  const int iu = j < 3 ? 0 : (j < 5 ? 1 : (j == 5 ? 2 : j - 6));
  const int iv = j < 3 ? j : (j < 5 ? j - 2 : (j == 5 ? 2 : 3));
  double a = 0., az = 0.;
  if (pre == LSA_MATCH_SUCCESS)
  {
#pragma unroll
    for (int s = 0; s < KMAX; ++s)
      if ((mask >> s) & 1u)
      {
        const float4 p = fs.cand[s][slot];
        const float u = iu == 0 ? p.x : (iu == 1 ? p.y : p.z);
        const float v = iv == 0 ? p.x : (iv == 1 ? p.y : (iv == 2 ? p.z : 1.f));
        a += (double)(u * v);  // float product, double sum (PCL 1.10 dense branch); x * 1.f is x
        az += (double)p.z;
      }
  }
  if (live && write && gl < 8)
  {
    fs.acc[j][slot] = a;
    if (gl == 0)
    {
      fs.acc[8][slot] = az;
      fs.nsel[slot] = nsel;
      fs.last_d2[slot] = last_d2;
      fs.pre[slot] = pre;
    }
  }
}

// The rest of the match, one lane per keypoint: mean and covariance from the sums, pcl::eigen33 in double, the validity
// tests, the residual record (KeypointsMatcher.cxx:149-186, 229-272, 315-345).  Returns the status.
template <int TYPE, int QB, int KC>
__device__ __forceinline__ int finish_model(const FitStage<QB, KC>& fs, int slot, const MatchConst& c, double* __restrict__ rec, int cap, int i)
{
  int st = fs.pre[slot];
  double w = 0.;
  if (st == LSA_MATCH_SUCCESS)
  {
    const float4 q4 = fs.q4[slot];
    CovAccum<double> acc;
    acc.a0 = fs.acc[0][slot]; acc.a1 = fs.acc[1][slot]; acc.a2 = fs.acc[2][slot]; acc.a3 = fs.acc[3][slot]; acc.a4 = fs.acc[4][slot];
    acc.a5 = fs.acc[5][slot]; acc.a6 = fs.acc[6][slot]; acc.a7 = fs.acc[7][slot]; acc.a8 = fs.acc[8][slot];
    Vec3<double> mean, e0, e1, e2;
    double l0 = 0, l1 = 0, l2 = 0;
    Sym3<double> cov;
    acc.finish(fs.nsel[slot], mean, cov);
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
          for (int b = 0; b < 3; ++b) A[a * 3 + b] = ((V[a][0] * d0) * V[b][0] + (V[a][1] * d1) * V[b][1]) + (V[a][2] * d2) * V[b][2];
        if (!isfinite(A[0]) || !isfinite(d0 * d1 * d2)) st = LSA_MATCH_INVALID_NUMERICAL;
        else w = 1.0;
      }
    }
    if (st == LSA_MATCH_SUCCESS) write_record(rec, cap, i, A, mean, (double)q4.x, (double)q4.y, (double)q4.z, w);
  }
  if (st != LSA_MATCH_SUCCESS) rec[(size_t)15 * cap + i] = 0.;  // Weights[i] = 0 for rejected keypoints
  return st;
}

// the whole target for one query, by the whole wavefront: best[0 .. k) uniform across it
template <int KMAX>
__device__ __forceinline__ void search_whole_target(const float4* __restrict__ sorted, int npoints, float qx, float qy, float qz, int k, knn_key (&best)[KMAX])
{
  const int lane = threadIdx.x & 63;
  LaneList<KMAX> L;
  L.reset();
  for (int c = lane; c < npoints; c += 64)
  {
    const float4 p = sorted[c];
    const float dx = qx - p.x, dy = qy - p.y, dz = qz - p.z;
    L.insert(make_key((dx * dx + dy * dy) + dz * dz, __float_as_int(p.w)));
  }
  merge_lists<KMAX, 64>(L, k, best);
}

// Pose and motion of the launch, into LDS: the launch's own arguments, or what the link / gate in front of it left on the
// device (enqueued ahead of its inputs: go == 1, or the iteration was called off and the workgroup returns false).  They
// always go through ONE copy in LDS, whichever source they have: a pointer that may lead into the kernel's arguments or
// elsewhere, and even a choice between the two value by value, makes the compiler move pose and motion (or all 1.4 KB of
// arguments) into scratch memory.  (The arguments themselves are never written: a kernel argument that is modified lives in
// scratch memory.)  One trip to memory: go travels with the block.
__device__ __forceinline__ bool load_inputs(const FusedArgs& a, IcpInputs& gin)
{
  __shared__ unsigned long long go;
  if (a.gate)
  {
    constexpr int words = (int)(sizeof(IcpInputs) / 8);
    static_assert(words + 1 <= 256 && offsetof(IcpGate, in) == 8, "one word per thread, go in front");
    const unsigned long long* src = reinterpret_cast<const unsigned long long*>(a.gate);
    if ((int)threadIdx.x < words) reinterpret_cast<unsigned long long*>(&gin)[threadIdx.x] = src[1 + threadIdx.x];
    else if ((int)threadIdx.x == words) go = src[0];
  }
  else if (threadIdx.x == 0)
  {
    go = 1ull;
    gin.pose = a.pose;
    if (a.undistort) gin.ic = a.ic;
  }
  __syncthreads();
  return go == 1ull;
}

// One workgroup's share of a type: 256 / G keypoints searched by G lanes each.  FUSE: the groups go on to the front
// part of the model fit (group_stage) and the workgroup's first wavefront finishes the 256 / G matches; otherwise the
// neighbour lists go to memory for the model kernel.
template <int KMAX, int G, int TYPE, bool FUSE, int KC>
__device__ __forceinline__ void search_type(const FusedArgs& a, IcpInputs& gin, const FusedType& t, int block, SearchShared& sh, FitStage<256 / G, KC>& fs)
{
  constexpr int QB = 256 / G;
  const int tid = threadIdx.x, gl = tid % G, ql = tid / G;
  const int q = block * QB + ql;
  const bool active = q < t.nq;
  // the keypoint first, on its way while the launch's inputs arrive: they do not depend on each other, and behind a link or a
  // gate the inputs are a trip to memory of their own
  float4 q4 = {0.f, 0.f, 0.f, 0.f}, b4 = {0.f, 0.f, 0.f, 0.f};
  if (active)
  {
    q4 = t.queries[2 * (size_t)q];
    if (a.undistort) b4 = t.queries[2 * (size_t)q + 1];
  }
  if (!load_inputs(a, gin)) return;
  const Rigid& pose = gin.pose;
  const InterpConst* const ic = a.undistort ? &gin.ic : nullptr;
  if (t.route_stats)
  {
    if (tid < 6) sh.route[tid] = 0;
    if (tid < 32) sh.phase[tid] = 0;
    __syncthreads();
  }
  float qx = 0.f, qy = 0.f, qz = 0.f;
  if (active)
  {
    // KeypointsMatcher: worldPoint = PosePrior * basePoint in double, narrowed to float for the search
    // (KeypointsMatcher.cxx:117-118, KDTreePCLAdaptor.h:96-100)
    auto place = [&](const Rigid& P, const InterpConst* C) {
      if (C)
      {
        // Slam::RefineUndistortion's step for this keypoint (k_undistort of lsa_transform.hip, same arithmetic): the lanes
        // of the group all work it out, the first one stores it for everything after this match
        Rigid U;
        interp_eval(*C, __hiloint2double(__float_as_int(b4.y), __float_as_int(b4.x)), U);
        double ux, uy, uz;
        rigid_apply(U, (double)q4.x, (double)q4.y, (double)q4.z, ux, uy, uz);
        q4.x = (float)ux; q4.y = (float)uy; q4.z = (float)uz;
        if (gl == 0) const_cast<float4*>(t.queries)[2 * (size_t)q] = q4;
      }
      double wx, wy, wz;
      rigid_apply(P, (double)q4.x, (double)q4.y, (double)q4.z, wx, wy, wz);
      qx = (float)wx; qy = (float)wy; qz = (float)wz;
    };
    place(pose, ic);
    if constexpr (FUSE) { if (gl == 0) fs.q4[ql] = q4; }
  }
  knn_key best[KMAX];
#pragma unroll
  for (int s = 0; s < KMAX; ++s) best[s] = kKeyEmpty;
  float ub = INFINITY;
  const unsigned long long tick0 = t.trace ? wall_clock64() : 0ull;
  const RowTable tb = {sh.b, sh.e, sh.p};
  const int outcome = group_search<KMAX, G>(t.desc, t.gp, qx, qy, qz, t.k, t.far_d2, active, gl, tid, tb, sh.save, best, ub, t.route_stats ? sh.route : nullptr, t.trace ? sh.phase : nullptr);
  int cnt = 0;
#pragma unroll
  for (int s = 0; s < KMAX; ++s)
    if (s < t.k && best[s] != kKeyEmpty) ++cnt;
  const unsigned long long tick1 = t.trace ? wall_clock64() : 0ull;
  if constexpr (FUSE)
  {
    // kStagePending: no block of the grid settles it -- the finishing wavefront searches the whole target for it
    const bool tail = active && outcome == kOutTail;
    if (tail && gl == 0) fs.pre[ql] = kStagePending;
    group_stage<KMAX, TYPE, G, QB, KC>(best, outcome == kOutFar ? kKnnFar : cnt, t.mc, t.xyzl, gl, ql, active && !tail, true, fs);
    if (t.trace && (tid & 63) == 0) sh.phase[(tid >> 6) * 8 + 4] = wall_clock64();
    __syncthreads();  // the workgroup's QB keypoints are staged (and the tables free)
    if (tid >= 64) return;
    if (t.trace && tid == 0) sh.phase[5] = wall_clock64();
    // the first wavefront finishes them, one lane each
    const int lane = tid;
    if (lane < LSA_MATCH_NSTATUS) fs.lh[lane] = 0;
    const int i = block * QB + lane;
    const bool have = lane < QB && i < t.nq;
    unsigned long long pending = __ballot(have && fs.pre[lane < QB ? lane : 0] == kStagePending);
    while (pending)
    {
      const int src = __ffsll((long long)pending) - 1;
      pending &= pending - 1;
      const float4 b4 = fs.q4[src];
      double wx, wy, wz;
      rigid_apply(pose, (double)b4.x, (double)b4.y, (double)b4.z, wx, wy, wz);
      search_whole_target<KMAX>(t.gp.sorted[0], t.npoints, (float)wx, (float)wy, (float)wz, t.k, best);
      int c2 = 0;
#pragma unroll
      for (int s = 0; s < KMAX; ++s)
        if (s < t.k && best[s] != kKeyEmpty) ++c2;
      // every group of the wavefront holds the same lists: all take part, the first one stores
      group_stage<KMAX, TYPE, G, QB, KC>(best, c2, t.mc, t.xyzl, gl, src, true, lane < G, fs);
      if (lane == 0) atomicAdd(t.hist + LSA_MATCH_NSTATUS, 1);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    if (have)
    {
#ifdef LSA_ABLATE_FINISH
      const int st = fs.pre[lane] == LSA_MATCH_SUCCESS ? LSA_MATCH_MSE_TOO_LARGE : fs.pre[lane];  // (timing experiment only)
#else
      const int st = finish_model<TYPE, QB, KC>(fs, lane, t.mc, t.rec, t.cap, i);
#endif
      t.status[i] = (uint8_t)st;
      atomicAdd(&fs.lh[st], 1);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // one wavefront: its LDS atomics are done, in order
    if (lane < LSA_MATCH_NSTATUS && fs.lh[lane]) atomicAdd(&t.hist[lane], fs.lh[lane]);
  }
  if (!FUSE && gl == 0 && active)
  {
#pragma unroll
    for (int s = 0; s < KMAX; ++s)
      if (s < t.k)
      {
        t.knn_idx[(size_t)s * t.cap + q] = key_idx(best[s]);
        t.knn_d2[(size_t)s * t.cap + q] = key_d2(best[s]);
      }
    // kPending: no block of the grid settles it -- the model kernel's wavefront searches the whole target for it
    t.knn_cnt[q] = outcome == kOutTail ? kPending : (outcome == kOutFar ? kKnnFar : cnt);
  }
  if (t.route_stats)
  {
    if (!FUSE) __syncthreads();  // (FUSE: only the first wavefront is left, behind the workgroup's barrier)
    if (t.trace && tid == 0)
    {
      const unsigned long long tick2 = wall_clock64();
      unsigned xcc = 0, hwid = 0;
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
      unsigned long long* tr = t.trace + (size_t)blockIdx.x * 12;
      tr[0] = tick0; tr[1] = tick1; tr[2] = tick2; tr[3] = ((unsigned long long)xcc << 32) | hwid;
      for (int i = 0; i < 6; ++i) tr[4 + i] = (unsigned long long)sh.route[i];
      // clocks of the phases, 16 bits each, relative to the start of the search
      // (of the wavefront that was staged last: the one the workgroup waited for)
      int w = 0;
      for (int v = 1; v < 4; ++v)
        if (sh.phase[v * 8 + 4] > sh.phase[w * 8 + 4]) w = v;
      auto rel = [&](int i) { return sh.phase[i] > tick0 ? ((sh.phase[i] - tick0) & 0xffffull) : 0ull; };
      tr[10] = rel(w * 8 + 0) | (rel(w * 8 + 1) << 16) | (rel(w * 8 + 2) << 32) | (rel(w * 8 + 3) << 48);
      tr[11] = rel(w * 8 + 4) | (rel(5) << 16);
    }
    if (tid < 6 && sh.route[tid])
    {
      if (tid == 5) atomicMax(&t.hist[LSA_MATCH_NSTATUS + 2 + tid], sh.route[tid]);
      else atomicAdd(&t.hist[LSA_MATCH_NSTATUS + 2 + tid], sh.route[tid]);
    }
  }
}

// lanes per query
constexpr int kGE = 8, kGP = 8, kGB = 8;

// The model fits of one ICP iteration as a launch of their own (two-launch form), one thread per keypoint, all types in
// one launch.  What the search could not settle inside the grid's blocks (kPending) is searched here first: the
// wavefront of such a keypoint walks the whole target for it, every lane keeping the k best of its share, one merge at
// the end.
template <int KE, int QPB>
struct ModelShared
{
  float4 nb[KE * QPB];  // edge candidates staged in LDS
  float nd[KE * QPB];
  int lh[LSA_MATCH_NSTATUS];
};

// QPB keypoints per workgroup, one thread each (the first QPB threads; every thread of the workgroup calls)
template <int KMAX, int TYPE, int QPB, typename SH>
__device__ __forceinline__ void model_type(const Rigid& pose, const FusedType& t, int block, SH& sh)
{
  const int tid = threadIdx.x, lane = tid & 63;
  if (tid < LSA_MATCH_NSTATUS) sh.lh[tid] = 0;
  __syncthreads();
  const int i = block * QPB + tid;
  const bool have = tid < QPB && i < t.nq;
  int n = 0;
  if (have && !t.mc.bad_param) n = t.knn_cnt[i];
  unsigned long long pending = __ballot(have && n == kPending);
  while (pending)
  {
    const int src = __ffsll((long long)pending) - 1;
    pending &= pending - 1;
    const int qi = block * QPB + (tid - lane) + src;
    const float4 q4 = t.queries[2 * (size_t)qi];
    double wx, wy, wz;
    rigid_apply(pose, (double)q4.x, (double)q4.y, (double)q4.z, wx, wy, wz);
    knn_key best[KMAX];
    search_whole_target<KMAX>(t.gp.sorted[0], t.npoints, (float)wx, (float)wy, (float)wz, t.k, best);
    if (lane == src)
    {
      int cnt = 0;
#pragma unroll
      for (int s = 0; s < KMAX; ++s)
        if (s < t.k)
        {
          t.knn_idx[(size_t)s * t.cap + qi] = key_idx(best[s]);
          t.knn_d2[(size_t)s * t.cap + qi] = key_d2(best[s]);
          if (best[s] != kKeyEmpty) ++cnt;
        }
      n = cnt;
      atomicAdd(t.hist + LSA_MATCH_NSTATUS, 1);
    }
  }
  if (have)
  {
    const int st = fit_model<KMAX, TYPE>(
      t.queries[2 * (size_t)i], t.mc, n, [&](int s) { return t.knn_idx[(size_t)s * t.cap + i]; }, [&](int s) { return t.knn_d2[(size_t)s * t.cap + i]; },
      t.xyzl, sh.nb, sh.nd, QPB, tid, t.rec, t.cap, i);
    t.status[i] = (uint8_t)st;
    atomicAdd(&sh.lh[st], 1);
  }
  __syncthreads();
  if (tid < LSA_MATCH_NSTATUS && sh.lh[tid]) atomicAdd(&t.hist[tid], sh.lh[tid]);
}

// The searches of one ICP iteration, all keypoint types in one launch -- and, when FUSE, the model fits of the same
// keypoints: the lanes that searched a keypoint gather its neighbours, filter them and sum up their covariance
// (group_stage: nothing of it leaves the chip), the workgroup's first wavefront does the eigen-decompositions and writes
// the records (finish_model).  No second launch, no neighbour lists in memory.
template <int KE, int KP, int KB, bool FUSE>
__global__ __launch_bounds__(256, 4) void k_search_all(FusedArgs a)
{
  constexpr int QB = 256 / kGE;
  constexpr int KC = FUSE ? (KE > KP ? (KE > KB ? KE : KB) : (KP > KB ? KP : KB)) : 1;
  static_assert(kGE == kGP && kGP == kGB, "one workgroup serves 256 / G keypoints of any type");
  __shared__ SearchShared sh;
  __shared__ FitStage<QB, KC> fs;
  __shared__ IcpInputs gin;  // pose and motion of the launch (load_inputs, from inside search_type: behind the keypoints' loads)
  // Hardware blocks b, b + 8, ... share an XCD (and its L2).  Every XCD gets one contiguous eighth of EVERY type's
  // blocks -- neighbouring keypoints (scan order) search a compact region of the target through one L2 -- and the
  // types with the longest searches (edges: no early out, larger k) first.
  const int xcd = blockIdx.x % 8, j = blockIdx.x / 8;
  const int se = (a.t[0].nblocks + 7) / 8, sp = (a.t[1].nblocks + 7) / 8;
  if (j < se)
  {
    const int block = xcd * se + j;
    if (block >= a.t[0].nblocks) return;
    search_type<KE, kGE, LSA_EDGE, FUSE, KC>(a, gin, a.t[0], block, sh, fs);
  }
  else if (j < se + sp)
  {
    const int block = xcd * sp + (j - se);
    if (block >= a.t[1].nblocks) return;
    search_type<KP, kGP, LSA_PLANE, FUSE, KC>(a, gin, a.t[1], block, sh, fs);
  }
  else
  {
    if constexpr (KB > 0)
    {
      const int sb = (a.t[2].nblocks + 7) / 8;
      const int block = xcd * sb + (j - se - sp);
      if (block >= a.t[2].nblocks) return;
      search_type<KB, kGB, LSA_BLOB, FUSE, KC>(a, gin, a.t[2], block, sh, fs);
    }
  }
}

template <int KE, int KP, int KB>
__global__ __launch_bounds__(kModelBlock) void k_model_all(FusedArgs a)
{
  __shared__ ModelShared<KE, kModelBlock> sh;
  if (a.gate && a.gate->go != 1ull) return;  // (only reached with a gate when a type's parameters are invalid: nothing is searched for it)
  int b = blockIdx.x;
  if (b < a.t[0].mblocks) { model_type<KE, LSA_EDGE, kModelBlock>(a.pose, a.t[0], b, sh); return; }
  b -= a.t[0].mblocks;
  if (b < a.t[1].mblocks) { model_type<KP, LSA_PLANE, kModelBlock>(a.pose, a.t[1], b, sh); return; }
  b -= a.t[1].mblocks;
  if constexpr (KB > 0) model_type<KB, LSA_BLOB, kModelBlock>(a.pose, a.t[2], b, sh);
}

template <int KE, int KP, int KB>
void launch_fused(lsa_ctx* ctx, const FusedArgs& a, double search_bytes, double model_bytes, hipStream_t st)
{
  const int grid = 8 * ((a.t[0].nblocks + 7) / 8 + (a.t[1].nblocks + 7) / 8 + (a.t[2].nblocks + 7) / 8);
  if (a.fuse_model)
  {
    // one launch: every workgroup fits the models of the keypoints it has searched
    ProfScope ps(ctx, "match_search_model", search_bytes + model_bytes, st);
    hipLaunchKernelGGL((k_search_all<KE, KP, KB, true>), dim3(grid), dim3(256), 0, st, a);
    return;
  }
  if (grid > 0)
  {
    ProfScope ps(ctx, "match_search", search_bytes, st);
    hipLaunchKernelGGL((k_search_all<KE, KP, KB, false>), dim3(grid), dim3(256), 0, st, a);
  }
  ProfScope ps(ctx, "match_model", model_bytes, st);
  hipLaunchKernelGGL((k_model_all<KE, KP, KB>), dim3(a.t[0].mblocks + a.t[1].mblocks + a.t[2].mblocks), dim3(kModelBlock), 0, st, a);
}

}  // namespace

namespace lsa
{

// Enqueues the matches `preps` describes (at most one per keypoint type, every one with a non-empty target and at
// least one keypoint) as two launches on `st`: the searches of all types, then their model fits.
int enqueue_fused_match(lsa_ctx* ctx, const MatchPrep* preps, int count, const double pose[16], hipStream_t st, const InterpConst* undistort, int gate, bool gate_undistorts)
{
  FusedArgs a;
  std::memset(&a, 0, sizeof(a));
  if (gate >= 0) a.gate = reinterpret_cast<const IcpGate*>(ctx->gate_dev + (size_t)gate * kGateWords);
  else row_major_to_rt(pose, a.pose.R, a.pose.t);
  int kmax[3] = {1, 1, 1};
  const int lanes[3] = {kGE, kGP, kGB};
  double search_bytes = 0, model_bytes = 0;
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
    t.knn_idx = mb.knn_idx; t.knn_d2 = mb.knn_d2; t.knn_cnt = mb.knn_cnt;
    t.rec = mb.rec; t.status = mb.status; t.cap = mb.cap;
    t.hist = p.hist;
    t.route_stats = ctx->route_stats ? 1 : 0;
    t.trace = ctx->route_stats ? reinterpret_cast<unsigned long long*>(ctx->trace_dev) : nullptr;
    kmax[p.type] = p.mc.k;
    // a type whose parameters are invalid is not searched (BAD_MODEL_PARAMETRIZATION for every keypoint)
    t.nblocks = p.mc.bad_param ? 0 : (int)(((size_t)p.nq * lanes[p.type] + 255) / 256);
    t.mblocks = (p.nq + kModelBlock - 1) / kModelBlock;
    // algorithmic bytes (B_icp = K (32 + 32 k + 136)): keypoint in, k gathered target points (32 B each in
    // the reference's layout), residual record + status out -- counted ONCE per keypoint for the one-launch form.  The
    // two-launch form splits the same total: the search takes the keypoint and the k points, the model fit the record.
    search_bytes += (double)p.nq * (32 + p.mc.k * 32);
    model_bytes += (double)p.nq * 136;
  }
  if (a.t[0].mblocks + a.t[1].mblocks + a.t[2].mblocks == 0) return LSA_OK;
  // the search kernel fits the models too when every type that has keypoints is searched (a type with invalid
  // parameters is not: its keypoints get their status from the model kernel)
  if (undistort)
  {
    // lsa_match_types_undistorted has checked that every keypoint of the set is searched by this launch
    a.undistort = 1;
    a.ic = *undistort;
  }
  if (gate >= 0 && gate_undistorts) a.undistort = 1;
  a.fuse_model = ctx->fused_model ? 1 : 0;
  for (int k = 0; k < 3; ++k)
    if (a.t[k].mblocks > 0 && a.t[k].nblocks == 0) a.fuse_model = 0;
  // the list lengths compiled in: the reference's defaults (edges 8 ego-motion / 10 localization, planes 5, no blobs)
  // get kernels of their own -- one kernel has one register budget, the longest list in it sets it for every type
  const int ke = kmax[0], kp = kmax[1];
  const bool blobs = a.t[2].nq > 0;
  if (!blobs && ke <= 8 && kp <= 5) launch_fused<8, 5, 0>(ctx, a, search_bytes, model_bytes, st);
  else if (!blobs && ke <= 10 && kp <= 5) launch_fused<10, 5, 0>(ctx, a, search_bytes, model_bytes, st);
  else if (kp <= 5) launch_fused<16, 5, 16>(ctx, a, search_bytes, model_bytes, st);
  else if (kp <= 8) launch_fused<16, 8, 16>(ctx, a, search_bytes, model_bytes, st);
  else launch_fused<16, 16, 16>(ctx, a, search_bytes, model_bytes, st);
  return LSA_OK;
}

}  // namespace lsa
