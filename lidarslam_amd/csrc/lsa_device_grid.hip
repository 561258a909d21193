// lsa_device_grid.hip -- LidarSlam::RollingGrid (slam_lib/include/LidarSlam/RollingGrid.h:63-212,
// slam_lib/src/RollingGrid.cxx) resident on the device: voxel insertion with the sampling modes, rolling,
// decay, bounding-box sub-map extraction straight into a kNN target.  SURVEY.md 8f-1.
//
// The reference keeps the map as unordered_map<outer voxel, unordered_map<leaf voxel, Voxel>>.  Here the map is ONE
// array of voxels sorted by the 64-bit key (outer index << 32 | leaf index) -- keys, points and counts in three
// parallel arrays, double-buffered:
//   Add      the batch is keyed and sorted by (key, arrival order) (runs sorted by a workgroup each, merged by rank); the
//            first thread of a run of equal keys finds the voxel by binary search and folds the run's points for it IN
//            ARRIVAL ORDER through the reference's per-point rule (FIRST / LAST / MAX_INTENSITY / CENTER_POINT, fixed
//            points, one count per Add call), exactly what the sequential loop of RollingGrid.cxx:183-312 does to that
//            voxel; new voxels are merged in by rank (two binary searches, one scatter) -- no hash table, no atomics on
//            voxels; seven launches for all the maps of a keyframe together
//   Roll     a shift of the outer coordinates keeps the key order: folded into Add's merge (on its own: transform + stable
//            compaction)
//   decay    ClearOldPoints: stable compaction
//   sub-map  voxels whose outer index lies in the box, stable compaction straight into the target's point buffer
// Iteration order.  The reference's Get / BuildSubMapKdTree hand the points out in libstdc++'s hash iteration order,
// an accident of the container; the device map hands them out in KEY ORDER (outer index, then leaf index as
// unsigned), and so do the oracle and the host RollingGrid when "OrderedMaps" is set (the default): a defined order
// in place of an accidental one, documented with the other deviations in DESIGN.md 4.3.
// Nothing here waits for the device except the calls that return a size or points to the host.
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <string.h>
#include "lsa_ctx.h"
#include "lsa_device_math.h"

using namespace lsa;

namespace
{
typedef unsigned long long u64;
constexpr u64 kNoKey = ~0ull;  // points outside the grid: sorted behind every voxel

struct GridParams
{
  int grid_size;
  float resolution;   // (float)VoxelResolution
  double resolution_d;
  float leaf;         // (float)LeafSize
  double leaf_d;
  int sampling;
  unsigned min_frames;
};
// state the kernels read and write (device memory, kStInts ints)
enum { kStN = 0, kStNbPoints = 1, kStUpdated = 2, kStPosX = 3, kStGroups = 6, kStNew = 7, kStOff = 8, kStSub = 11, kStTmp = 12 /* 6 ints */, kStSubFirst = 18, kStCompact = 19, kStPred = 20 /* 6 ints: lo[3], hi[3]: outer voxels of the box a sub-map was extracted ahead for */, kStInts = 32 };

__device__ __forceinline__ float ordered_to_float(unsigned u) { return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u); }

__device__ __forceinline__ int round_to_int(float v)
{
  // Eigen's .round().cast<int>(): round half away from zero, then a C cast (out of range: INT_MIN, as on x86-64)
  const float r = roundf(v);
  return (r >= -2147483648.f && r < 2147483648.f) ? (int)r : (int)0x80000000;
}

// ---- stable compaction: chunk counts -> scatter (the predicate is evaluated twice) ---------------------------------------
// Two launches: every scatter block sums the counts of the chunks before it itself (a map of a few hundred thousand voxels
// is a few hundred chunks), the last block leaves the total.
template <typename Pred>
__global__ __launch_bounds__(256) void k_compact_count(Pred pred, const int* __restrict__ n_ptr, int n_fixed, int* __restrict__ chunk_count)
{
  __shared__ int cnt[4];
  const int n = n_ptr ? *n_ptr : n_fixed;
  if (blockIdx.x * 1024 >= n) { if (threadIdx.x == 0) chunk_count[blockIdx.x] = 0; return; }
  int mine = 0;
  for (int q = 0; q < 4; ++q)
  {
    const int i = blockIdx.x * 1024 + q * 256 + threadIdx.x;
    if (i < n && pred(i)) ++mine;
  }
  for (int o = 32; o > 0; o >>= 1) mine += __shfl_down(mine, o);
  if ((threadIdx.x & 63) == 0) cnt[threadIdx.x >> 6] = mine;
  __syncthreads();
  if (threadIdx.x == 0) chunk_count[blockIdx.x] = cnt[0] + cnt[1] + cnt[2] + cnt[3];
}
// total_out: where the number kept goes (never the word n_ptr points at: the other blocks still read that one);
// base_ptr: the output starts behind *base_ptr elements (appending), at 0 when null
template <typename Pred, typename Emit>
__global__ __launch_bounds__(256) void k_compact_scatter(Pred pred, Emit emit, const int* __restrict__ n_ptr, int n_fixed, const int* __restrict__ chunk_count,
                                                         const int* __restrict__ base_ptr, int* __restrict__ total_out,
                                                         u64* __restrict__ host_out = nullptr, unsigned host_tag = 0, int* __restrict__ clear_flag = nullptr)
{
  __shared__ int wave_cnt[4];
  __shared__ int before[4];
  const int n = n_ptr ? *n_ptr : n_fixed;
  const bool last = blockIdx.x == gridDim.x - 1;
  if (blockIdx.x * 1024 >= n && !last) return;
  int mine = 0;
  for (int c = threadIdx.x; c < (int)blockIdx.x; c += 256) mine += chunk_count[c];
  for (int o = 32; o > 0; o >>= 1) mine += __shfl_down(mine, o);
  if ((threadIdx.x & 63) == 0) before[threadIdx.x >> 6] = mine;
  __syncthreads();
  int run = (base_ptr ? *base_ptr : 0) + before[0] + before[1] + before[2] + before[3];
  for (int q = 0; q < 4; ++q)
  {
    const int i = blockIdx.x * 1024 + q * 256 + threadIdx.x;
    const bool keep = i < n && pred(i);
    const u64 ballot = __ballot(keep);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    __syncthreads();  // wave_cnt of the round before has been read
    if (lane == 0) wave_cnt[wv] = __popcll(ballot);
    __syncthreads();
    int base = run;
    for (int w = 0; w < wv; ++w) base += wave_cnt[w];
    if (keep) emit(i, base + __popcll(ballot & ((1ull << lane) - 1ull)));
    run += wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
  }
  if (last && threadIdx.x == 0)
  {
    *total_out = run;
    if (clear_flag) *clear_flag = 0;
    // the total for the host: tag and count in ONE 8-byte store into coherent host memory (no copy, no event)
    if (host_out) __hip_atomic_store(host_out, ((u64)host_tag << 32) | (u64)(unsigned)run, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

struct MapView
{
  u64* keys;
  float4* pts;      // two float4 per voxel point
  unsigned* count;
};

// ---- Roll (RollingGrid.cxx:117-157) --------------------------------------------------------------------------------
// bounding box of the batch: ordered-int atomics into st[kStTmp .. +5]
__device__ __forceinline__ int f2o_i(float f) { const int i = __float_as_int(f); return i >= 0 ? i : i ^ 0x7fffffff; }
__device__ __forceinline__ float o2f_i(int i) { return __int_as_float(i >= 0 ? i : i ^ 0x7fffffff); }
__device__ __forceinline__ void d_batch_bbox(int bx, const float4* __restrict__ batch, int n, int* __restrict__ st)
{
  __shared__ float smn[4][3], smx[4][3];
  const int i = bx * blockDim.x + threadIdx.x;
  float mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
  if (i < n)
  {
    const float4 a = batch[2 * (size_t)i];
    mn[0] = mx[0] = a.x; mn[1] = mx[1] = a.y; mn[2] = mx[2] = a.z;
  }
  for (int d = 0; d < 3; ++d)
    for (int o = 32; o > 0; o >>= 1)
    {
      mn[d] = fminf(mn[d], __shfl_down(mn[d], o));
      mx[d] = fmaxf(mx[d], __shfl_down(mx[d], o));
    }
  if ((threadIdx.x & 63) == 0)
    for (int d = 0; d < 3; ++d) { smn[threadIdx.x >> 6][d] = mn[d]; smx[threadIdx.x >> 6][d] = mx[d]; }
  __syncthreads();
  // six atomics per workgroup (every wavefront aiming at the same six words was 15 us for 27 k points)
  if (threadIdx.x < 3)
  {
    const int d = threadIdx.x;
    atomicMin(&st[kStTmp + d], f2o_i(fminf(fminf(smn[0][d], smn[1][d]), fminf(smn[2][d], smn[3][d]))));
    atomicMax(&st[kStTmp + 3 + d], f2o_i(fmaxf(fmaxf(smx[0][d], smx[1][d]), fmaxf(smx[2][d], smx[3][d]))));
  }
}
// how many outer voxels the grid has to move so that the box fits (one thread); explicit box: roll_to != nullptr
__global__ void k_roll_decide(GridParams p, int* __restrict__ st, int use_box)
{
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  const double halfGridSize = static_cast<double>(p.grid_size) / 2 * p.resolution_d;
  const float h = (float)halfGridSize;
  for (int d = 0; d < 3; ++d)
  {
    int off = 0;
    if (use_box)
    {
      const float mnv = o2f_i(st[kStTmp + d]), mxv = o2f_i(st[kStTmp + 3 + d]);
      const float pos = __int_as_float(st[kStPosX + d]);
      const float down = mnv - (pos - h);
      const float up = mxv - (pos + h);
      float o = (up + down) / 2.f;
      const float lo = fminf(down, 0.f), hi = fmaxf(up, 0.f);
      o = fminf(fmaxf(o, lo), hi);
      off = round_to_int(o / p.resolution);
    }
    st[kStOff + d] = off;
    st[kStPosX + d] = __float_as_int(__int_as_float(st[kStPosX + d]) + (float)off * p.resolution);
  }
  // re-arm the box for the next batch
  for (int d = 0; d < 3; ++d) { st[kStTmp + d] = 0x7fffffff; st[kStTmp + 3 + d] = (int)0x80000000; }
}
struct RollPred
{
  const u64* keys;
  const int* st;
  int grid_size;
  __device__ bool shifted(int i, u64& out) const
  {
    const u64 k = keys[i];
    int id = (int)(unsigned)(k >> 32);
    const int g = grid_size;
    int z = id / (g * g);
    id -= z * g * g;
    int y = id / g;
    int x = id - y * g;
    x -= st[kStOff + 0]; y -= st[kStOff + 1]; z -= st[kStOff + 2];
    if (x < 0 || y < 0 || z < 0 || x >= g || y >= g || z >= g) return false;
    out = ((u64)(unsigned)(z * g * g + y * g + x) << 32) | (k & 0xffffffffull);
    return true;
  }
  __device__ bool operator()(int i) const { u64 o; return shifted(i, o); }
};
struct RollEmit
{
  RollPred pred;
  MapView src, dst;
  __device__ void operator()(int i, int at) const
  {
    u64 k;
    pred.shifted(i, k);
    dst.keys[at] = k;
    dst.pts[2 * (size_t)at] = src.pts[2 * (size_t)i];
    dst.pts[2 * (size_t)at + 1] = src.pts[2 * (size_t)i + 1];
    dst.count[at] = src.count[i];
  }
};
__global__ void k_after_roll(int* __restrict__ st)
{
  if (threadIdx.x == 0 && blockIdx.x == 0)
  {
    st[kStN] = st[kStCompact];  // what the compaction kept
    if (st[kStOff] | st[kStOff + 1] | st[kStOff + 2]) st[kStNbPoints] = st[kStCompact];  // Roll recounts the points when the grid moved (RollingGrid.cxx:136-137, 155)
  }
}

__device__ __forceinline__ int lower_bound_u64(const u64* __restrict__ a, int n, u64 key)
{
  int lo = 0, hi = n;
  while (lo < hi)
  {
    const int mid = (lo + hi) >> 1;
    if (a[mid] < key) lo = mid + 1; else hi = mid;
  }
  return lo;
}

// ---- Add (RollingGrid.cxx:160-318) in seven launches --------------------------------------------------------------------
// (The first version -- bounding box, roll decision, roll compaction, keys, library radix sort, heads, fold, compaction of
// the new voxels, merge, state -- was twenty-five dependent launches, and the next localization waits for the last of
// them.)  Seven: box -> keys of the batch + survivors of the roll counted -> runs of 4096 sorted in LDS -> runs merged by
// rank -> fold per voxel (straight off the sorted batch) -> map = surviving old voxels (re-keyed) merged with the new ones
// by rank -> state.  Nothing is decided in a launch of its own: every kernel works the roll's shift out for itself from the
// committed grid position and the batch's box, and the last kernel commits position and counts.
struct Shift
{
  int off[3];     // outer voxels the grid moves by (Roll, RollingGrid.cxx:117-157)
  float pos[3];   // grid position after the move
  bool any;
};
__device__ __forceinline__ Shift roll_shift(const GridParams& p, const int* __restrict__ st, int use_box)
{
  Shift s;
  const double halfGridSize = static_cast<double>(p.grid_size) / 2 * p.resolution_d;
  const float h = (float)halfGridSize;
  s.any = false;
#pragma unroll
  for (int d = 0; d < 3; ++d)
  {
    const float pos = __int_as_float(st[kStPosX + d]);
    int off = 0;
    if (use_box)
    {
      const float mnv = o2f_i(st[kStTmp + d]), mxv = o2f_i(st[kStTmp + 3 + d]);
      const float down = mnv - (pos - h);
      const float up = mxv - (pos + h);
      float o = (up + down) / 2.f;
      const float lo = fminf(down, 0.f), hi = fmaxf(up, 0.f);
      o = fminf(fmaxf(o, lo), hi);
      off = round_to_int(o / p.resolution);
    }
    s.off[d] = off;
    s.pos[d] = pos + (float)off * p.resolution;
    s.any = s.any || off != 0;
  }
  return s;
}
// A voxel's place in the order of the map, comparable between the grid before and after a move: (z, y, x) of the outer
// voxel in the coordinates BEFORE the move (biased, 21 bits each: a voxel that is about to enter the grid has coordinates
// outside of it), then the leaf index.  Without a move the outer index itself does.
struct VKey
{
  u64 hi;
  unsigned lo;
};
__device__ __forceinline__ bool vless(const VKey& a, const VKey& b) { return a.hi < b.hi || (a.hi == b.hi && a.lo < b.lo); }
__device__ __forceinline__ u64 biased3(int x, int y, int z)
{
  auto c = [](int v) { const int lim = (1 << 20) - 1; return (u64)(unsigned)((v < -lim ? -lim : (v > lim ? lim : v)) + (1 << 20)); };
  return (c(z) << 42) | (c(y) << 21) | c(x);
}
__device__ __forceinline__ VKey vkey_of_old(u64 key, bool any, int g)
{
  VKey k;
  k.lo = (unsigned)(key & 0xffffffffull);
  int id = (int)(unsigned)(key >> 32);
  if (!any) { k.hi = (u64)(unsigned)id; return k; }
  const int z = id / (g * g); id -= z * g * g;
  const int y = id / g; const int x = id - y * g;
  k.hi = biased3(x, y, z);
  return k;
}
// the key of a voxel of the grid AFTER the move, in that order
__device__ __forceinline__ VKey vkey_of_new(u64 key, const Shift& s, int g)
{
  VKey k;
  k.lo = (unsigned)(key & 0xffffffffull);
  int id = (int)(unsigned)(key >> 32);
  if (!s.any) { k.hi = (u64)(unsigned)id; return k; }
  const int z = id / (g * g); id -= z * g * g;
  const int y = id / g; const int x = id - y * g;
  k.hi = biased3(x + s.off[0], y + s.off[1], z + s.off[2]);
  return k;
}
__device__ __forceinline__ int lower_bound_old(const u64* __restrict__ keys, int n, const VKey& t, bool any, int g)
{
  int lo = 0, hi = n;
  while (lo < hi)
  {
    const int mid = (lo + hi) >> 1;
    if (vless(vkey_of_old(keys[mid], any, g), t)) lo = mid + 1; else hi = mid;
  }
  return lo;
}
// The same within [lo, hi) -- the answer is known to lie in [lo, hi].
__device__ __forceinline__ int lower_bound_old_in(const u64* __restrict__ keys, int lo, int hi, const VKey& t, bool any, int g)
{
  while (lo < hi)
  {
    const int mid = (lo + hi) >> 1;
    if (vless(vkey_of_old(keys[mid], any, g), t)) lo = mid + 1; else hi = mid;
  }
  return lo;
}
// ... and by a whole wavefront for ONE key (the same in every lane): 64 probes per round trip instead of one, three or four
// dependent loads for a map of a million voxels instead of twenty.  Every lane returns the answer.
__device__ __forceinline__ int lower_bound_old_wave(const u64* __restrict__ keys, int n, const VKey& t, bool any, int g)
{
  const int lane = threadIdx.x & 63;
  int lo = 0, hi = n;  // the answer lies in [lo, hi]
  while (hi - lo > 64)
  {
    const int step = (hi - lo + 64) / 65;  // >= 1; probes at lo + step * (lane + 1) - 1, clamped: non-decreasing along the lanes
    const int pos = min(hi - 1, lo + step * (lane + 1) - 1);
    const bool less = vless(vkey_of_old(keys[pos], any, g), t);
    const int c = __popcll(__ballot(less));  // the keys ascend: the probes below the target are the first c lanes'
    // the answer is beyond probe c - 1 and not beyond probe c
    const int nlo = c == 0 ? lo : min(hi - 1, lo + step * c - 1) + 1;
    const int nhi = c == 64 ? hi : min(hi - 1, lo + step * (c + 1) - 1);
    lo = nlo; hi = nhi;
  }
  const int pos = lo + lane;
  const bool less = pos < hi && vless(vkey_of_old(keys[pos], any, g), t);
  return lo + __popcll(__ballot(less));
}
// does the voxel survive the move, and under which key
__device__ __forceinline__ bool shifted_key(u64 k, const Shift& s, int g, u64& out)
{
  int id = (int)(unsigned)(k >> 32);
  int z = id / (g * g);
  id -= z * g * g;
  int y = id / g;
  int x = id - y * g;
  x -= s.off[0]; y -= s.off[1]; z -= s.off[2];
  if (x < 0 || y < 0 || z < 0 || x >= g || y >= g || z >= g) return false;
  out = ((u64)(unsigned)(z * g * g + y * g + x) << 32) | (k & 0xffffffffull);
  return true;
}

// launch 2: blocks [0, kblocks): the keys of the batch in the grid after the move; the others: 1024 old voxels each, which
// of them survive the move -- every voxel's rank among the survivors of its chunk, and the chunk's count
__device__ __forceinline__ void d_add_keys(int bx, const float4* __restrict__ batch, int n, int kblocks, GridParams p, const int* __restrict__ st, int use_box,
                                                  u64* __restrict__ keys, const u64* __restrict__ old_keys, int* __restrict__ old_local,
                                                  int* __restrict__ old_chunks, int ochunks)
{
  const Shift s = roll_shift(p, st, use_box);
  const int g = p.grid_size;
  if ((int)bx < kblocks)
  {
    const int i = bx * 256 + threadIdx.x;
    if (i >= n) return;
    const float4 a = batch[2 * (size_t)i];
    const float pt[3] = {a.x, a.y, a.z};
    int out[3], in[3];
    bool inside = true;
#pragma unroll
    for (int d = 0; d < 3; ++d)
    {
      // voxelGridOrigin = VoxelGridPosition - int(GridSize / 2) * VoxelResolution (:177)
      const float origin = s.pos[d] - (float)((double)(g / 2) * p.resolution_d);
      out[d] = round_to_int((pt[d] - origin) / p.resolution);
      inside = inside && out[d] >= 0 && out[d] < g;
      const float center = (float)out[d] * p.resolution + origin;
      in[d] = round_to_int((pt[d] - center) / p.leaf);
    }
    const unsigned idx_out = (unsigned)(out[2] * g * g + out[1] * g + out[0]);
    const unsigned idx_in = (unsigned)(in[2] * g * g + in[1] * g + in[0]);  // possibly "negative": the reference's own index (:200-202)
    keys[i] = inside ? (((u64)idx_out << 32) | idx_in) : kNoKey;
    return;
  }
  __shared__ int wave_cnt[4];
  const int chunk = bx - kblocks;
  if (chunk >= ochunks) return;  // (a launch shared with a bigger map)
  const int N = st[kStN];
  int run = 0;
  for (int q = 0; q < 4; ++q)
  {
    const int i = chunk * 1024 + q * 256 + threadIdx.x;
    u64 nk;
    const bool keep = i < N && shifted_key(old_keys[i], s, g, nk);
    const u64 ballot = __ballot(keep);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) wave_cnt[wv] = __popcll(ballot);
    __syncthreads();
    int base = run;
    for (int w = 0; w < wv; ++w) base += wave_cnt[w];
    if (i < N) old_local[i] = base + __popcll(ballot & ((1ull << lane) - 1ull));
    run += wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
  }
  if (threadIdx.x == 0) old_chunks[chunk] = run;
}

// launch 3: runs of 4096 (key, arrival index) pairs sorted by one workgroup (bitonic; the pairs are unique, so the order
// is the stable order by key).  A thread holds four consecutive pairs in registers: of the 78 steps of the network, the
// 23 whose partner is one of the thread's own pairs are done in place, the 45 whose partner sits in another lane of the
// wavefront go through lane exchanges, and only the 10 that cross wavefronts go through LDS (two buffers, one barrier
// each).  (The first version did all 78 through LDS with a barrier each: 55 us a run.)
constexpr int kRun = 4096;
struct SortPair
{
  u64 k;
  unsigned i;
};
__device__ __forceinline__ bool pair_gt(const SortPair& a, const SortPair& b) { return a.k > b.k || (a.k == b.k && a.i > b.i); }
// the value of lane (lane ^ M) for one 32-bit word: DPP operands and gfx950's permlane swaps, no LDS crossbar
template <int M>
__device__ __forceinline__ unsigned word_xor(unsigned x, int lane)
{
  if (M == 1) return (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0xB1, 0xF, 0xF, true);   // quad_perm [1, 0, 3, 2]
  if (M == 2) return (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x4E, 0xF, 0xF, true);   // quad_perm [2, 3, 0, 1]
  if (M == 4)
  {
    const unsigned up = (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x104, 0xF, 0xF, true);  // row_shl:4: lane i <- i + 4
    const unsigned dn = (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xF, 0xF, true);  // row_shr:4: lane i <- i - 4
    return (lane & 4) ? dn : up;
  }
  if (M == 8) return (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x128, 0xF, 0xF, true);   // row_ror:8
  if (M == 16)
  {
    const auto r = __builtin_amdgcn_permlane16_swap(x, x, false, false);  // [0]: rows 0 0 2 2, [1]: rows 1 1 3 3
    return (lane & 16) ? r[0] : r[1];
  }
  const auto r = __builtin_amdgcn_permlane32_swap(x, x, false, false);    // [0]: lower half twice, [1]: upper half twice
  return (lane & 32) ? r[0] : r[1];
}
template <int M>
__device__ __forceinline__ SortPair lane_xor(const SortPair& v, int lane)
{
  SortPair o;
  const unsigned lo = word_xor<M>((unsigned)(v.k & 0xffffffffull), lane), hi = word_xor<M>((unsigned)(v.k >> 32), lane);
  o.k = ((u64)hi << 32) | lo;
  o.i = word_xor<M>(v.i, lane);
  return o;
}
__device__ __forceinline__ void d_sort_runs(int bx, const u64* keys, int n, u64* out_keys, unsigned* __restrict__ out_idx)  // keys == out_keys: in place, run by run
{
  __shared__ u64 sk[2][kRun];
  __shared__ unsigned si[2][kRun];
  const int base = bx * kRun, t = threadIdx.x;
  if (base >= n) return;  // (a launch shared with a bigger batch)
  SortPair v[4];
#pragma unroll
  for (int e = 0; e < 4; ++e)
  {
    const int i = base + 4 * t + e;
    v[e].k = i < n ? keys[i] : kNoKey;
    v[e].i = i < n ? (unsigned)i : 0xffffffffu;
  }
  int buf = 0;
  for (int k = 2; k <= kRun; k <<= 1)
    for (int j = k >> 1; j > 0; j >>= 1)
    {
      if (j <= 2)
      {
        // partner = another pair of this thread (constant register indices: j = 1 pairs 0-1 and 2-3, j = 2 pairs 0-2 and 1-3)
        auto cx = [&](SortPair& a, SortPair& b, int pos) {
          const bool up = ((pos & k) == 0);
          if (pair_gt(a, b) == up) { const SortPair x = a; a = b; b = x; }
        };
        if (j == 1) { cx(v[0], v[1], 4 * t); cx(v[2], v[3], 4 * t + 2); }
        else { cx(v[0], v[2], 4 * t); cx(v[1], v[3], 4 * t + 1); }
      }
      else
      {
        const int tj = j >> 2;                  // the partner thread is t ^ tj, same place inside the thread
        const bool lower = (t & tj) == 0;
        SortPair o[4];
        if (tj < 64)
        {
          const int lane = t & 63;
          // one of six code paths, chosen by a scalar branch (the step is the same for the whole workgroup)
          switch (__builtin_amdgcn_readfirstlane(tj))
          {
            case 1: _Pragma("unroll") for (int e = 0; e < 4; ++e) o[e] = lane_xor<1>(v[e], lane); break;
            case 2: _Pragma("unroll") for (int e = 0; e < 4; ++e) o[e] = lane_xor<2>(v[e], lane); break;
            case 4: _Pragma("unroll") for (int e = 0; e < 4; ++e) o[e] = lane_xor<4>(v[e], lane); break;
            case 8: _Pragma("unroll") for (int e = 0; e < 4; ++e) o[e] = lane_xor<8>(v[e], lane); break;
            case 16: _Pragma("unroll") for (int e = 0; e < 4; ++e) o[e] = lane_xor<16>(v[e], lane); break;
            default: _Pragma("unroll") for (int e = 0; e < 4; ++e) o[e] = lane_xor<32>(v[e], lane); break;
          }
        }
        else
        {
#pragma unroll
          for (int e = 0; e < 4; ++e) { sk[buf][4 * t + e] = v[e].k; si[buf][4 * t + e] = v[e].i; }
          __syncthreads();
          const int pt = t ^ tj;
#pragma unroll
          for (int e = 0; e < 4; ++e) { o[e].k = sk[buf][4 * pt + e]; o[e].i = si[buf][4 * pt + e]; }
          buf ^= 1;  // the next step through LDS writes the other buffer: this one may still be read
        }
#pragma unroll
        for (int e = 0; e < 4; ++e)
        {
          const bool up = (((4 * t + e) & k) == 0);
          const bool take_min = lower == up;
          const bool mine_gt = pair_gt(v[e], o[e]);
          if (mine_gt == take_min) v[e] = o[e];
        }
      }
    }
#pragma unroll
  for (int e = 0; e < 4; ++e)
  {
    const int i = base + 4 * t + e;
    if (i < n) { out_keys[i] = v[e].k; out_idx[i] = v[e].i; }
  }
}
// launch 4 (more than one run): every pair's place is the number of pairs of all runs in front of it
__device__ __forceinline__ void d_merge_runs(int bx, const u64* __restrict__ keys, const unsigned* __restrict__ idx, int n, u64* __restrict__ out_keys,
                                                    unsigned* __restrict__ out_idx)
{
  const int e = bx * 256 + threadIdx.x;
  if (e >= n) return;
  const u64 k = keys[e];
  const unsigned id = idx[e];
  const int mine = e / kRun;
  int rank = e - mine * kRun;
  for (int q = 0; q * kRun < n; ++q)
  {
    if (q == mine) continue;
    int lo = q * kRun, hi = min(n, (q + 1) * kRun);
    const int first = lo;
    while (lo < hi)
    {
      const int mid = (lo + hi) >> 1;
      const u64 km = keys[mid];
      if (km < k || (km == k && idx[mid] < id)) lo = mid + 1; else hi = mid;
    }
    rank += lo - first;
  }
  out_keys[rank] = k;
  out_idx[rank] = id;
}

// launch 5: one thread per place of the sorted batch; the first of a run of equal keys folds the run's points into the
// voxel, in arrival order, by the reference's per-point rule.  An existing voxel is updated where it is (the old array);
// a new one is left at the thread's own place in `fresh`, flagged, with its rank among the new ones of the block.
__device__ __forceinline__ void d_add_fold(int bx, const float4* __restrict__ batch, int n, const u64* __restrict__ skeys, const unsigned* __restrict__ sorder,
                                                  GridParams p, int* __restrict__ st, int use_box, MapView map, MapView fresh, int* __restrict__ fresh_flag,
                                                  int* __restrict__ fresh_chunks, int fixed, double time, int* __restrict__ vrank = nullptr, bool only_flags = false)
{
  __shared__ int wave_cnt[4];
  if (bx * 256 >= n) return;  // (a launch shared with a bigger batch)
  const Shift sft = roll_shift(p, st, use_box);
  const int g = p.grid_size;
  const int j0 = bx * 256 + threadIdx.x;
  const u64 key = j0 < n ? skeys[j0] : kNoKey;
  const bool head = j0 < n && key != kNoKey && (j0 == 0 || skeys[j0 - 1] != key);
  bool is_fresh = false;
  // Where the block's 256 sorted keys lie in the old array: the places of its first and of its last valid key, found by a
  // wavefront each (64 probes per round trip); every thread then searches between the two -- a few hundred voxels, a
  // handful of cache lines the block shares -- instead of the whole map.
  __shared__ int bound[2];
  const int N = st[kStN];
  {
    const int wv = threadIdx.x >> 6;
    if (wv < 2)
    {
      // the last valid key of the block: the keys ascend and the invalid ones (kNoKey) sort last
      int jl = min(n, bx * 256 + 256) - 1;
      const u64 kf = skeys[bx * 256];
      u64 kl = skeys[jl];
      int res = wv == 0 ? 0 : N;
      if (wv == 0 && kf != kNoKey) res = lower_bound_old_wave(map.keys, N, vkey_of_new(kf, sft, g), sft.any, g);
      if (wv == 1 && kl != kNoKey) res = lower_bound_old_wave(map.keys, N, vkey_of_new(kl, sft, g), sft.any, g);
      if ((threadIdx.x & 63) == 0) bound[wv] = res;
    }
    __syncthreads();
  }
  if (head)
  {
    const VKey target = vkey_of_new(key, sft, g);
    const int at = lower_bound_old_in(map.keys, bound[0], min(N, max(bound[1], bound[0])), target, sft.any, g);
    bool exists = false;
    if (at < N)
    {
      const VKey k = vkey_of_old(map.keys[at], sft.any, g);
      exists = k.hi == target.hi && k.lo == target.lo;
    }
    float4 va, vb;      // the voxel's point
    unsigned count = 0;
    bool have = exists;
    bool changed = false;
    if (exists) { va = map.pts[2 * (size_t)at]; vb = map.pts[2 * (size_t)at + 1]; count = map.count[at]; }
    else { va = make_float4(0.f, 0.f, 0.f, 0.f); vb = va; }
    // CENTER_POINT (:253): the centre is that of the leaf voxel of the point at hand, voxelGridCenterIn - VoxelResolution / 2.f
    // + LeafSize * voxelCoordIn -- two leaf voxels of one outer voxel can share an inner index (To1d of coordinates around
    // zero), so the points of one run do not all have the same centre
    float base[3] = {0.f, 0.f, 0.f};  // voxelGridCenterIn: the same for the whole run
    if (p.sampling == 3)
    {
      int id = (int)(unsigned)(key >> 32);
      const int oz = id / (g * g); id -= oz * g * g;
      const int oy = id / g; const int ox = id - oy * g;
      const int out[3] = {ox, oy, oz};
#pragma unroll
      for (int d = 0; d < 3; ++d)
      {
        const float origin = sft.pos[d] - (float)((double)(g / 2) * p.resolution_d);
        base[d] = (float)out[d] * p.resolution + origin;
      }
    }
    bool counted = false;
    // CENTROID (:263-297).  The reference keeps, per voxel that an earlier point of this Add fell into, the running mean of
    // those points -- and, INSIDE its loop over the points, pulls EVERY such voxel's point towards its mean once per point
    // of the whole cloud that gets as far as the end of the loop body ((point * count + mean) / (count + 1), :282-297).  A
    // voxel's point therefore depends on how many such points lie between and behind its own in arrival order: vrank.
    float mean[3] = {0.f, 0.f, 0.f};
    unsigned mean_count = 0;
    bool in_mean = false;
    int prev_rank = -1;
    auto pull = [&](int times) {
      const float c = (float)count, c1 = (float)(count + 1);
      for (int it = 0; it < times; ++it)
      {
        const float nx = (va.x * c + mean[0]) / c1, ny = (va.y * c + mean[1]) / c1, nz = (va.z * c + mean[2]) / c1;
        if (nx == va.x && ny == va.y && nz == va.z) break;  // a fixed point of the step: nothing moves any more
        va.x = nx; va.y = ny; va.z = nz;
      }
    };
    for (int j = j0; j < n && skeys[j] == key; ++j)
    {
      const unsigned src = sorder[j];
      const float4 a = batch[2 * (size_t)src], b = batch[2 * (size_t)src + 1];
      if (only_flags)
      {
        // (first of the CENTROID launches) does this point get to the end of the loop body?  Not when its voxel holds a
        // fixed point (:219-220) -- from before, or because an earlier point of this very call made it one
        bool through = true;
        if (!have) { have = true; vb.w = __uint_as_float((fixed ? 1u : 0u) << 24); }
        else if (((__float_as_uint(vb.w) >> 24) & 0xffu) == 1) through = false;
        else vb.w = __uint_as_float((__float_as_uint(vb.w) & 0x00ffffffu) | ((fixed ? 1u : 0u) << 24));
        vrank[src] = through ? 1 : 0;
        continue;
      }
      if (!have)
      {
        va = a; vb = b; have = true; changed = true;  // new voxel: the point as it is (:206-212)
      }
      else
      {
        const unsigned label = (__float_as_uint(vb.w) >> 24) & 0xffu;
        if (label == 1) continue;  // the voxel holds a fixed point: nothing of this point is taken, not even its time (:219-220)
        if (p.sampling == 4)
        {
          // the pulls of the points of other voxels since this voxel's last one, then this point into the mean
          if (in_mean) pull(vrank[src] - prev_rank - 1);
          const float mc = (float)mean_count, mc1 = (float)(mean_count + 1);
          mean[0] = (mean[0] * mc + a.x) / mc1; mean[1] = (mean[1] * mc + a.y) / mc1; mean[2] = (mean[2] * mc + a.z) / mc1;
          ++mean_count;
          in_mean = true;
        }
        if (p.sampling == 1) { va = a; vb = b; changed = true; }                       // LAST
        else if (p.sampling == 2) { if (b.z > vb.z) { va = a; vb = b; changed = true; } }  // MAX_INTENSITY
        else if (p.sampling == 3)
        {
          const float pt[3] = {a.x, a.y, a.z};
          float centre[3];
#pragma unroll
          for (int d = 0; d < 3; ++d) centre[d] = base[d] - p.resolution / 2.f + p.leaf * (float)round_to_int((pt[d] - base[d]) / p.leaf);
          const float d1x = a.x - centre[0], d1y = a.y - centre[1], d1z = a.z - centre[2];
          const float d0x = va.x - centre[0], d0y = va.y - centre[1], d0z = va.z - centre[2];
          // Eigen's Vector3f norm: sqrt(x^2 + (y^2 + z^2))
          if (sqrtf(d1x * d1x + (d1y * d1y + d1z * d1z)) < sqrtf(d0x * d0x + (d0y * d0y + d0z * d0z))) { va = a; vb = b; changed = true; }
        }
      }
      if (p.sampling == 4)
      {
        if (in_mean) pull(1);  // this point's own turn of the loop at :282-297
        prev_rank = vrank[src];
      }
      // voxel.point.time = currentTime; label = fixed (:300-306); one count per Add call (:307-311)
      const long long tb = __double_as_longlong(time);
      vb.x = __int_as_float((int)(tb & 0xffffffffll));
      vb.y = __int_as_float((int)(tb >> 32));
      vb.w = __uint_as_float((__float_as_uint(vb.w) & 0x00ffffffu) | ((fixed ? 1u : 0u) << 24));
      if (!counted) { ++count; counted = true; }
    }
    if (!only_flags)
    {
    if (p.sampling == 4 && in_mean) pull(vrank[n] - prev_rank - 1);  // the points of the cloud behind this voxel's last one
    if (exists)
    {
      map.pts[2 * (size_t)at] = va;
      map.pts[2 * (size_t)at + 1] = vb;
      map.count[at] = count;
    }
    else
    {
      is_fresh = true;
      fresh.keys[j0] = key;
      fresh.pts[2 * (size_t)j0] = va;
      fresh.pts[2 * (size_t)j0 + 1] = vb;
      fresh.count[j0] = count;
    }
    if (changed) st[kStUpdated] = 1;
    }
  }
  if (only_flags)
  {
    // points outside the grid (no key) never enter the loop body
    if (j0 < n && key == kNoKey) vrank[sorder[j0]] = 0;
    return;
  }
  // rank of every place among the block's new voxels (the places that hold none get the rank the next one would)
  const u64 ballot = __ballot(is_fresh);
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  if (lane == 0) wave_cnt[wv] = __popcll(ballot);
  __syncthreads();
  int before = 0;
  for (int w = 0; w < wv; ++w) before += wave_cnt[w];
  if (j0 < n) fresh_flag[j0] = ((before + __popcll(ballot & ((1ull << lane) - 1ull))) << 1) | (is_fresh ? 1 : 0);
  if (threadIdx.x == 0) fresh_chunks[bx] = wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
}

// launch 6: the new map.  Blocks [0, oblocks): 1024 old voxels each -- a survivor's place is its rank among the survivors
// plus the number of new voxels in front of it; the other blocks: 256 places of the sorted batch each -- a new voxel's
// place is its rank among the new ones plus the number of survivors in front of it.  Every block scans the chunk counts
// of both arrays for itself (dynamic LDS: ochunks + fchunks + 2 ints).
__device__ __forceinline__ void d_add_merge(int bx, GridParams p, int* __restrict__ st, int use_box, MapView old, const int* __restrict__ old_local,
                                                   const int* __restrict__ old_chunks, int ochunks, int oblocks, const u64* __restrict__ skeys, int n, MapView fresh,
                                                   const int* __restrict__ fresh_flag, const int* __restrict__ fresh_chunks, int fchunks, MapView dst)
{
  extern __shared__ int scan[];  // [ochunks + 1] exclusive scan of the survivors per chunk, then [fchunks + 1] of the new voxels per block
  __shared__ int carry;
  // oblocks: where the launch's blocks for the sorted batch begin (it may be shared with a bigger map)
  if (bx < oblocks ? bx >= ochunks : (bx - oblocks) * 256 >= n) return;
  int* const oscan = scan;
  int* const fscan = scan + ochunks + 1;
  const Shift sft = roll_shift(p, st, use_box);
  const int g = p.grid_size;
  const int N = st[kStN];
  // both scans, 256 entries at a time
  for (int which = 0; which < 2; ++which)
  {
    const int* src = which ? fresh_chunks : old_chunks;
    int* out = which ? fscan : oscan;
    const int cnt = which ? fchunks : ochunks;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (int b0 = 0; b0 < cnt; b0 += 256)
    {
      const int c = b0 + threadIdx.x;
      const int v = c < cnt ? src[c] : 0;
      int inc = v;
      for (int o = 1; o < 64; o <<= 1)
      {
        const int t = __shfl_up(inc, o);
        if ((threadIdx.x & 63) >= o) inc += t;
      }
      __shared__ int wsum[4];
      if ((threadIdx.x & 63) == 63) wsum[threadIdx.x >> 6] = inc;
      __syncthreads();
      int add = carry;
      for (int w = 0; w < (int)(threadIdx.x >> 6); ++w) add += wsum[w];
      if (c < cnt) out[c] = add + inc - v;
      __syncthreads();
      if (threadIdx.x == 255) carry = add + inc;
      __syncthreads();
    }
    if (threadIdx.x == 0) out[cnt] = carry;
    __syncthreads();
  }
  const int survivors = oscan[ochunks], created = fscan[fchunks];
  if (bx == 0 && threadIdx.x == 0) { st[kStCompact] = survivors; st[kStNew] = created; }
  auto fresh_before = [&](int j) { return j >= n ? created : fscan[j >> 8] + (fresh_flag[j] >> 1); };
  auto survivors_before = [&](int i) { return i >= N ? survivors : oscan[i >> 10] + old_local[i]; };
  if ((int)bx < oblocks)
  {
    for (int q = 0; q < 4; ++q)
    {
      const int i = bx * 1024 + q * 256 + threadIdx.x;
      if (i >= N) continue;
      u64 nk;
      if (!shifted_key(old.keys[i], sft, g, nk)) continue;
      const int at = survivors_before(i) + fresh_before(lower_bound_u64(skeys, n, nk));
      dst.keys[at] = nk;
      dst.pts[2 * (size_t)at] = old.pts[2 * (size_t)i];
      dst.pts[2 * (size_t)at + 1] = old.pts[2 * (size_t)i + 1];
      dst.count[at] = old.count[i];
    }
    return;
  }
  const int j = (bx - oblocks) * 256 + threadIdx.x;
  if (j >= n || !(fresh_flag[j] & 1)) return;
  const u64 key = skeys[j];
  const int at = fresh_before(j) + survivors_before(lower_bound_old(old.keys, N, vkey_of_new(key, sft, g), sft.any, g));
  dst.keys[at] = key;
  dst.pts[2 * (size_t)at] = fresh.pts[2 * (size_t)j];
  dst.pts[2 * (size_t)at + 1] = fresh.pts[2 * (size_t)j + 1];
  dst.count[at] = fresh.count[j];
}
// launch 7: the move and the counts become the grid's state (Roll recounts the points when the grid moved, :155)
__device__ __forceinline__ void d_add_commit(int bx, GridParams p, int* __restrict__ st, int use_box)
{
  if (threadIdx.x != 0 || bx != 0) return;
  const Shift s = roll_shift(p, st, use_box);
  const int survivors = st[kStCompact], created = st[kStNew];
  st[kStNbPoints] = (s.any ? survivors : st[kStNbPoints]) + created;
  st[kStN] = survivors + created;
  for (int d = 0; d < 3; ++d)
  {
    st[kStOff + d] = s.off[d];
    st[kStPosX + d] = __float_as_int(s.pos[d]);
    st[kStTmp + d] = 0x7fffffff;            // the box is re-armed for the next batch
    st[kStTmp + 3 + d] = (int)0x80000000;
  }
}

// One launch of each step serves all the maps of a keyframe (blockIdx.y = map): the insertions of the keypoint types run
// side by side instead of one behind the other on the stream they share.
struct AddOne
{
  const float4* batch;
  int n, use_box, fixed;
  double time;
  GridParams p;
  int* st;
  u64 *bkeys, *skeys;
  unsigned *border, *sorder;
  MapView map, fresh, dst;
  int *old_local, *old_chunks, *fresh_flag, *fresh_chunks;
  int* vrank;
  int ochunks;
};
struct AddBatch
{
  AddOne a[3];
  int kblocks, oblocks;  // of the launch: the largest of the maps'
};
__global__ __launch_bounds__(256) void k_batch_bbox(AddBatch b) { const AddOne& A = b.a[blockIdx.y]; if (A.use_box) d_batch_bbox(blockIdx.x, A.batch, A.n, A.st); }
__global__ __launch_bounds__(256) void k_add_keys(AddBatch b)
{
  const AddOne& A = b.a[blockIdx.y];
  d_add_keys(blockIdx.x, A.batch, A.n, b.kblocks, A.p, A.st, A.use_box, A.bkeys, A.map.keys, A.old_local, A.old_chunks, A.ochunks);
}
__global__ __launch_bounds__(1024) void k_sort_runs(AddBatch b) { const AddOne& A = b.a[blockIdx.y]; d_sort_runs(blockIdx.x, A.bkeys, A.n, A.bkeys, A.border); }
__global__ __launch_bounds__(256) void k_merge_runs(AddBatch b) { const AddOne& A = b.a[blockIdx.y]; d_merge_runs(blockIdx.x, A.bkeys, A.border, A.n, A.skeys, A.sorder); }
__global__ __launch_bounds__(256) void k_add_fold(AddBatch b)
{
  const AddOne& A = b.a[blockIdx.y];
  d_add_fold(blockIdx.x, A.batch, A.n, A.skeys, A.sorder, A.p, A.st, A.use_box, A.map, A.fresh, A.fresh_flag, A.fresh_chunks, A.fixed, A.time, A.vrank, false);
}
// CENTROID sampling only, in front of the fold: which points of the batch get to the end of the loop body (k_add_flags: the
// fold's own walk over the runs, nothing written but the flags), and how many of them lie in front of every point in
// ARRIVAL order (k_add_vscan: exclusive scan in place, one workgroup per map; [n] = all of them)
__global__ __launch_bounds__(256) void k_add_flags(AddBatch b)
{
  const AddOne& A = b.a[blockIdx.y];
  if (A.p.sampling != 4) return;
  d_add_fold(blockIdx.x, A.batch, A.n, A.skeys, A.sorder, A.p, A.st, A.use_box, A.map, A.fresh, A.fresh_flag, A.fresh_chunks, A.fixed, A.time, A.vrank, true);
}
__global__ __launch_bounds__(1024) void k_add_vscan(AddBatch b)
{
  const AddOne& A = b.a[blockIdx.y];
  if (A.p.sampling != 4) return;
  __shared__ int s[1024];
  int run = 0;
  for (int base = 0; base < A.n; base += 1024)
  {
    const int i = base + threadIdx.x;
    const int v = i < A.n ? A.vrank[i] : 0;
    s[threadIdx.x] = v;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1)
    {
      const int a = threadIdx.x >= (unsigned)o ? s[threadIdx.x - o] : 0;
      __syncthreads();
      s[threadIdx.x] += a;
      __syncthreads();
    }
    if (i < A.n) A.vrank[i] = run + s[threadIdx.x] - v;
    run += s[1023];
    __syncthreads();
  }
  if (threadIdx.x == 0) A.vrank[A.n] = run;
}
__global__ __launch_bounds__(256) void k_add_merge(AddBatch b)
{
  const AddOne& A = b.a[blockIdx.y];
  d_add_merge(blockIdx.x, A.p, A.st, A.use_box, A.map, A.old_local, A.old_chunks, A.ochunks, b.oblocks, A.skeys, A.n, A.fresh, A.fresh_flag, A.fresh_chunks, (A.n + 255) / 256,
              A.dst);
}
__global__ void k_add_commit(AddBatch b) { const AddOne& A = b.a[blockIdx.y]; d_add_commit(blockIdx.x, A.p, A.st, A.use_box); }

// ---- ClearOldPoints (RollingGrid.cxx:325-351) ----------------------------------------------------------------------
struct DecayPred
{
  const float4* pts;
  double now, threshold;
  __device__ bool operator()(int i) const
  {
    const float4 b = pts[2 * (size_t)i + 1];
    const unsigned label = (__float_as_uint(b.w) >> 24) & 0xffu;
    const double t = __hiloint2double(__float_as_int(b.y), __float_as_int(b.x));
    return !(!label && now - t > threshold);
  }
};
struct CopyEmit
{
  MapView src, dst;
  __device__ void operator()(int i, int at) const
  {
    dst.keys[at] = src.keys[i];
    dst.pts[2 * (size_t)at] = src.pts[2 * (size_t)i];
    dst.pts[2 * (size_t)at + 1] = src.pts[2 * (size_t)i + 1];
    dst.count[at] = src.count[i];
  }
};

// ---- Get / BuildSubMapKdTree (RollingGrid.cxx:95-114, 353-442) ---------------------------------------------------------
// the outer voxels the box [mn, mx] touches (:365-370): PositionToVoxel of both corners against the grid position the
// device holds, clamped to the grid.  The box comes from the caller (floats) or from the bounding-box words the context's
// lsa_keypoint_bboxes_begin left on the device (ordered unsigned, 6 per keypoint type).  Every thread works it out for
// itself (a handful of operations against a launch of its own).
struct BoxArg { float mn[3], mx[3]; };
struct SubMapPred
{
  const u64* keys;
  const float4* pts;
  const unsigned* count;
  const int* st;
  BoxArg box;
  const unsigned* ctx_box;
  const int* range;  // non-null: lo[3], hi[3] in outer voxels, worked out before (the box of a sub-map extracted ahead)
  int grid_size;
  float resolution;
  double resolution_d;
  int mode;          // 0 every voxel in the box; 1 count >= min_frames or fixed; 2 the others (count < min_frames and not fixed), only if pass 1 was short
  unsigned min_frames;
  int min_points;
  int boxed;         // 0: the whole map (Get / BuildSubMapKdTree()), 3: count > min_frames (Get(clean))
  __device__ bool operator()(int i) const
  {
    if (boxed == 0) return true;
    if (boxed == 3) return count[i] > min_frames;
    int id = (int)(unsigned)(keys[i] >> 32);
    const int g = grid_size;
    const int z = id / (g * g); id -= z * g * g;
    const int y = id / g; const int x = id - y * g;
    const int c[3] = {x, y, z};
    if (range)
    {
#pragma unroll
      for (int d = 0; d < 3; ++d)
        if (c[d] < range[d] || c[d] > range[3 + d]) return false;
    }
    else
#pragma unroll
    for (int d = 0; d < 3; ++d)
    {
      const float lo_f = ctx_box ? ordered_to_float(ctx_box[d]) : box.mn[d];
      const float hi_f = ctx_box ? ordered_to_float(ctx_box[3 + d]) : box.mx[d];
      const float origin = __int_as_float(st[kStPosX + d]) - (float)((double)(g / 2) * resolution_d);
      const int lo = round_to_int((lo_f - origin) / resolution), hi = round_to_int((hi_f - origin) / resolution);
      if (c[d] < (lo > 0 ? lo : 0) || c[d] > (hi < g - 1 ? hi : g - 1)) return false;
    }
    if (mode == 0) return true;
    const unsigned label = (__float_as_uint(pts[2 * (size_t)i + 1].w) >> 24) & 0xffu;
    if (mode == 1) return count[i] >= min_frames || label == 1;
    return st[kStSubFirst] < min_points && count[i] < min_frames && label != 1;  // st[kStSubFirst]: what pass 1 kept
  }
};
struct PointEmit
{
  const float4* pts;
  float4* out;
  __device__ void operator()(int i, int at) const
  {
    out[2 * (size_t)at] = pts[2 * (size_t)i];
    out[2 * (size_t)at + 1] = pts[2 * (size_t)i + 1];
  }
};
__global__ void k_set_int(int* __restrict__ p, int v) { if (threadIdx.x == 0 && blockIdx.x == 0) *p = v; }
__global__ void k_copy_int(int* __restrict__ dst, const int* __restrict__ src) { if (threadIdx.x == 0 && blockIdx.x == 0) *dst = *src; }
// the outer voxels the box of `words` (ordered unsigned, lsa_keypoint_bboxes_begin) touches: lo[3], hi[3]
__device__ __forceinline__ void box_voxels(const unsigned* __restrict__ words, int grid_size, float resolution, double resolution_d, const int* __restrict__ st, int lo[3],
                                           int hi[3])
{
#pragma unroll
  for (int d = 0; d < 3; ++d)
  {
    const float origin = __int_as_float(st[kStPosX + d]) - (float)((double)(grid_size / 2) * resolution_d);
    const int a = round_to_int((ordered_to_float(words[d]) - origin) / resolution), b = round_to_int((ordered_to_float(words[3 + d]) - origin) / resolution);
    lo[d] = a > 0 ? a : 0;
    hi[d] = b < grid_size - 1 ? b : grid_size - 1;
  }
}
// sub-map ahead of time: the voxel range of the PREDICTED box is kept in the state ...
__global__ void k_pred_box(const unsigned* __restrict__ words, int grid_size, float resolution, double resolution_d, int* __restrict__ st)
{
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  int lo[3], hi[3];
  box_voxels(words, grid_size, resolution, resolution_d, st, lo, hi);
  for (int d = 0; d < 3; ++d) { st[kStPred + d] = lo[d]; st[kStPred + 3 + d] = hi[d]; }
}
// ... and compared with that of the ACTUAL box when the localization asks: the sub-map only depends on the range of
// outer voxels the box touches (RollingGrid.cxx:363-442).  {tag, same} goes to the host in one 8-byte store.
__global__ void k_box_check(const unsigned* __restrict__ words, int grid_size, float resolution, double resolution_d, int* __restrict__ st, u64* __restrict__ host_out,
                            unsigned tag)
{
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  int lo[3], hi[3];
  box_voxels(words, grid_size, resolution, resolution_d, st, lo, hi);
  bool same = true;
  for (int d = 0; d < 3; ++d) same = same && lo[d] == st[kStPred + d] && hi[d] == st[kStPred + 3 + d];
  if (same) st[kStUpdated] = 0;  // the sub-map that is about to be taken over is of the map as it is now
  __hip_atomic_store(host_out, ((u64)tag << 32) | (same ? 1u : 0u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

}  // namespace

struct lsa_device_grid
{
  lsa_ctx* ctx = nullptr;
  // parameters (RollingGrid.h:170-212)
  int GridSize = 50;
  double VoxelResolution = 10.;
  double LeafSize = 0.2;
  unsigned MinFramesPerVoxel = 0;
  int Sampling = 2;  // MAX_INTENSITY
  double DecayingThreshold = -1.;
  // the map
  MapView buf[2] = {};
  int cur = 0;
  int cap = 0;
  int n_upper = 0;  // upper bound of the number of voxels (what has been added so far)
  int* st = nullptr;           // device state (16 ints)
  int* host_st = nullptr;      // pinned copy of it, refreshed behind every modification
  hipEvent_t ev_state = nullptr;
  // The grid's kernels run on a stream beside the context's (by default the context's look-ahead stream, see
  // lsa_device_grid_create): a keyframe goes into the map beside the next frame's work on the context's stream (and may
  // be enqueued by another host thread).  Where the two meet -- keypoints read, a target or
  // the scratch buffer written -- events order them: ev_in (context -> grid) before, ev_out (grid -> context) after.
  hipStream_t stream = nullptr;
  bool own_stream = false, shared_stream = false;
  hipEvent_t ev_in = nullptr, ev_out = nullptr, ev_sub = nullptr, ev_ahead = nullptr;
  u64* host_sub = nullptr;     // coherent host memory: {tag, size} of the sub-map being built, one 8-byte store by the kernel
  unsigned sub_tag = 0;
  int sub_target = -1;         // target index (slot * 3 + type) of the sub-map between _begin and _end
  bool sub_pending = false;    // kernels of a sub-map are on their way
  // a sub-map extracted AHEAD of time for a predicted box, into the context's spare map target (target[9 + type])
  u64* host_ahead = nullptr;   // coherent host memory: [0] {tag, size} of the extraction, [1] {tag, same box?} of the check
  unsigned ahead_tag = 0;
  int ahead_phase = 0;         // 0 none, 1 extraction on its way, 2 search grid on its way / ready
  int ahead_type = -1, ahead_min = 0, ahead_m = 0;
  bool take_pending = false;   // a comparison of _take_begin is on its way
  int take_slot = 0;
  int staged = 0;              // keypoints staged in `batch` by lsa_device_grid_stage_keypoints
  bool submap_valid = false;
  int submap_count = 0;
  // batch scratch
  int bcap = 0;
  float4* batch = nullptr;
  u64 *bkeys = nullptr, *skeys = nullptr;
  unsigned *border = nullptr, *sorder = nullptr;
  int *heads = nullptr, *fresh_flag = nullptr, *chunks = nullptr;
  int* vrank = nullptr;  // CENTROID sampling: how many points of the batch that take part in the loop body lie in front of every point (arrival order), [n] = all
  int* old_local = nullptr;    // [cap] rank of an old voxel among the survivors of its chunk (Add)
  MapView fresh = {};
  int chunk_cap = 0;
};

namespace
{
#define G_HIP(call)                                                                                    \
  do                                                                                                   \
  {                                                                                                    \
    hipError_t e__ = (call);                                                                           \
    if (e__ != hipSuccess) return g->ctx->fail(LSA_E_HIP, std::string(#call) + ": " + hipGetErrorString(e__)); \
  } while (0)

// n_upper counts every point ever added; the exact number of voxels comes back with the state copy that follows each
// modification, and replaces the bound as soon as the last of those copies has landed
void tighten(lsa_device_grid* g)
{
  if (g->n_upper > 0 && hipEventQuery(g->ev_state) == hipSuccess) g->n_upper = std::min(g->n_upper, std::max(g->host_st[kStN], 0));
}

// a sub-map extraction (on the context's stream) reads the map and uses the grid's scratch: modifications come behind it
int after_submap(lsa_device_grid* g)
{
  G_HIP(hipStreamWaitEvent(g->stream, g->ev_sub, 0));
  g->ahead_phase = 0;  // a sub-map extracted ahead of time was of the map before this modification
  return LSA_OK;
}

int order_after_context(lsa_device_grid* g)
{
  G_HIP(hipEventRecord(g->ev_in, g->ctx->stream));
  G_HIP(hipStreamWaitEvent(g->stream, g->ev_in, 0));
  return LSA_OK;
}
int order_context_after(lsa_device_grid* g)
{
  G_HIP(hipEventRecord(g->ev_out, g->stream));
  G_HIP(hipStreamWaitEvent(g->ctx->stream, g->ev_out, 0));
  return LSA_OK;
}

GridParams params_of(const lsa_device_grid* g)
{
  GridParams p;
  p.grid_size = g->GridSize;
  p.resolution = (float)g->VoxelResolution;
  p.resolution_d = g->VoxelResolution;
  p.leaf = (float)g->LeafSize;
  p.leaf_d = g->LeafSize;
  p.sampling = g->Sampling;
  p.min_frames = g->MinFramesPerVoxel;
  return p;
}

int alloc_view(lsa_device_grid* g, MapView& v, int cap)
{
  G_HIP(hipMalloc((void**)&v.keys, (size_t)cap * sizeof(u64)));
  G_HIP(hipMalloc((void**)&v.pts, (size_t)cap * 2 * sizeof(float4)));
  G_HIP(hipMalloc((void**)&v.count, (size_t)cap * sizeof(unsigned)));
  return LSA_OK;
}
void free_view(MapView& v)
{
  if (v.keys) (void)hipFree(v.keys);
  if (v.pts) (void)hipFree(v.pts);
  if (v.count) (void)hipFree(v.count);
  v = MapView{};
}
// an outgrown view while the grid lives: retired with its context (lsa_ctx.h: grave_dev), freed at the next frame's start
void retire_view(lsa_device_grid* g, MapView& v)
{
  retire_dev(g->ctx, v.keys);
  retire_dev(g->ctx, v.pts);
  retire_dev(g->ctx, v.count);
  v = MapView{};
}

// room for `want` voxels in both buffers of the map (contents kept) and for the chunk counters of a compaction over them
int ensure_map(lsa_device_grid* g, int want)
{
  if (want > g->cap)
  {
    const int cap = std::max(2 * want, 1 << 19);  // 46 MB for both buffers: growth (a device-wide stall) is rare
    G_HIP(hipStreamSynchronize(g->stream));
    MapView nb[2];
    for (int b = 0; b < 2; ++b)
    {
      int rc = alloc_view(g, nb[b], cap);
      if (rc) return rc;
    }
    if (g->cap > 0)
    {
      const MapView& o = g->buf[g->cur];
      G_HIP(hipMemcpy(nb[0].keys, o.keys, (size_t)g->cap * sizeof(u64), hipMemcpyDeviceToDevice));
      G_HIP(hipMemcpy(nb[0].pts, o.pts, (size_t)g->cap * 2 * sizeof(float4), hipMemcpyDeviceToDevice));
      G_HIP(hipMemcpy(nb[0].count, o.count, (size_t)g->cap * sizeof(unsigned), hipMemcpyDeviceToDevice));
      retire_view(g, g->buf[0]);
      retire_view(g, g->buf[1]);
    }
    g->buf[0] = nb[0];
    g->buf[1] = nb[1];
    g->cur = 0;
    g->cap = cap;
    retire_dev(g->ctx, g->old_local);
    g->old_local = nullptr;
    G_HIP(hipMalloc((void**)&g->old_local, (size_t)cap * sizeof(int)));
  }
  const int nchunks = (std::max(g->cap, g->bcap) + 1023) / 1024 + 1;
  if (nchunks > g->chunk_cap)
  {
    retire_dev(g->ctx, g->chunks);
    g->chunks = nullptr;
    G_HIP(hipMalloc((void**)&g->chunks, (size_t)nchunks * sizeof(int)));
    g->chunk_cap = nchunks;
  }
  return LSA_OK;
}

int ensure_batch(lsa_device_grid* g, int n)
{
  if (n <= g->bcap) return LSA_OK;
  // (twice what is asked for: outgrowing the batch retires eight buffers, and freeing them at the next frame's start waits for
  //  the device eight times -- 0.2-0.6 ms; a keyframe's keypoint count wanders by a quarter over the first hundred frames)
  const int cap = std::max(2 * n, 1 << 15);
  auto fr = [g](void* p) { retire_dev(g->ctx, p); };
  fr(g->batch); fr(g->bkeys); fr(g->skeys); fr(g->border); fr(g->sorder); fr(g->heads); fr(g->fresh_flag); fr(g->vrank);
  retire_view(g, g->fresh);
  G_HIP(hipMalloc((void**)&g->batch, (size_t)cap * 2 * sizeof(float4)));
  G_HIP(hipMalloc((void**)&g->bkeys, (size_t)cap * sizeof(u64)));
  G_HIP(hipMalloc((void**)&g->skeys, (size_t)cap * sizeof(u64)));
  G_HIP(hipMalloc((void**)&g->border, (size_t)cap * sizeof(unsigned)));
  G_HIP(hipMalloc((void**)&g->sorder, (size_t)cap * sizeof(unsigned)));
  G_HIP(hipMalloc((void**)&g->heads, (size_t)cap * sizeof(int)));
  G_HIP(hipMalloc((void**)&g->fresh_flag, (size_t)cap * sizeof(int)));
  G_HIP(hipMalloc((void**)&g->vrank, ((size_t)cap + 1) * sizeof(int)));
  int rc = alloc_view(g, g->fresh, cap);
  if (rc) return rc;
  g->bcap = cap;
  return LSA_OK;
}

// stable compaction of [0, n) (n on the device when n_ptr is given) by pred, emit(i, position); the number kept lands in
// *total (added to what is there when `append`)
template <typename Pred, typename Emit>
void compact(lsa_device_grid* g, Pred pred, Emit emit, const int* n_ptr, int n_bound, int* total, bool append = false, bool copy_back = true,
             hipStream_t on = nullptr, u64* host_out = nullptr, unsigned host_tag = 0, int* clear_flag = nullptr)
{
  hipStream_t st = on ? on : g->stream;
  const int nchunks = std::max((n_bound + 1023) / 1024, 1);
  hipLaunchKernelGGL((k_compact_count<Pred>), dim3(nchunks), dim3(256), 0, st, pred, n_ptr, n_bound, g->chunks);
  // compacting in place of the count it reads (Roll, ClearOldPoints), or appending behind it: the blocks of the
  // scatter still read the old count, the new one waits in a slot of its own until they are through
  const bool aside = n_ptr == total || append;
  int* const sum = aside ? g->st + kStCompact : total;
  hipLaunchKernelGGL((k_compact_scatter<Pred, Emit>), dim3(nchunks), dim3(256), 0, st, pred, emit, n_ptr, n_bound, g->chunks, append ? total : (const int*)nullptr, sum,
                     host_out, host_tag, clear_flag);
  if (aside && copy_back) hipLaunchKernelGGL(k_copy_int, dim3(1), dim3(64), 0, st, total, sum);  // otherwise the caller's next kernel takes it from st[kStCompact]
}

// the host's copy of the state follows every modification (asynchronously)
int refresh_state(lsa_device_grid* g)
{
  G_HIP(hipMemcpyAsync(g->host_st, g->st, kStInts * sizeof(int), hipMemcpyDeviceToHost, g->stream));
  G_HIP(hipEventRecord(g->ev_state, g->stream));
  G_HIP(hipEventRecord(g->ev_out, g->stream));
  return LSA_OK;
}

// Roll (always a pass into the other buffer: the host does not know whether the grid moves)
int roll(lsa_device_grid* g, bool use_box)
{
  hipStream_t st = g->stream;
  const GridParams p = params_of(g);
  hipLaunchKernelGGL(k_roll_decide, dim3(1), dim3(64), 0, st, p, g->st, use_box ? 1 : 0);
  const MapView src = g->buf[g->cur], dst = g->buf[1 - g->cur];
  RollPred pred{src.keys, g->st, g->GridSize};
  RollEmit emit{pred, src, dst};
  compact(g, pred, emit, g->st + kStN, std::max(g->n_upper, 1), g->st + kStN, false, false);
  hipLaunchKernelGGL(k_after_roll, dim3(1), dim3(64), 0, st, g->st);
  g->cur = 1 - g->cur;
  return LSA_OK;
}

// Add of the ns[i] points in gs[i]->batch (device), for up to three maps of one context at a time: seven launches
// whatever the number of maps
int add_batches(lsa_device_grid* const* gs, const int* ns, int count, bool fixed, double time, bool do_roll)
{
  lsa_device_grid* g = gs[0];  // (for the error macro; all maps share the context and the stream)
  hipStream_t st = g->stream;
  AddBatch b{};
  int kmax = 0, omax = 0, runs = 0;
  for (int i = 0; i < count; ++i)
  {
    lsa_device_grid* gi = gs[i];
    if (gi->ctx != g->ctx || gi->stream != st) return g->ctx->fail(LSA_E_ARG, "lsa_device_grid: the maps of one insertion share a context");
    tighten(gi);
    int rc = after_submap(gi);
    if (rc) return rc;
    rc = ensure_map(gi, gi->n_upper + ns[i]);
    if (rc) return rc;
    AddOne& A = b.a[i];
    A.batch = gi->batch; A.n = ns[i]; A.use_box = do_roll ? 1 : 0; A.fixed = fixed ? 1 : 0; A.time = time;
    A.p = params_of(gi); A.st = gi->st;
    A.bkeys = gi->bkeys; A.skeys = gi->skeys; A.border = gi->border; A.sorder = gi->sorder;
    A.map = gi->buf[gi->cur]; A.dst = gi->buf[1 - gi->cur]; A.fresh = gi->fresh;
    A.old_local = gi->old_local; A.old_chunks = gi->chunks; A.fresh_flag = gi->fresh_flag; A.fresh_chunks = gi->heads; A.vrank = gi->vrank;
    A.ochunks = std::max((gi->n_upper + 1023) / 1024, 1);
    kmax = std::max(kmax, (ns[i] + 255) / 256);
    omax = std::max(omax, A.ochunks);
    runs = std::max(runs, (ns[i] + kRun - 1) / kRun);
  }
  b.kblocks = kmax;
  b.oblocks = omax;
  const size_t lds = (size_t)(omax + kmax + 2) * sizeof(int);
  if (lds > 48 * 1024) return g->ctx->fail(LSA_E_CAPACITY, "lsa_device_grid: more than twelve million voxels in a map");
  double bytes = 0;
  for (int i = 0; i < count; ++i) bytes += (double)ns[i] * (32 + 12 + 44) + (double)gs[i]->n_upper * 44 * 2;
  {
    ProfScope ps(g->ctx, "map_add", bytes, st);
    const unsigned y = (unsigned)count;
    if (do_roll) hipLaunchKernelGGL(k_batch_bbox, dim3(kmax, y), dim3(256), 0, st, b);
    hipLaunchKernelGGL(k_add_keys, dim3(kmax + omax, y), dim3(256), 0, st, b);
    hipLaunchKernelGGL(k_sort_runs, dim3(runs, y), dim3(1024), 0, st, b);
    hipLaunchKernelGGL(k_merge_runs, dim3(kmax, y), dim3(256), 0, st, b);
    bool centroid = false;
    for (int i = 0; i < count; ++i) centroid = centroid || b.a[i].p.sampling == 4;
    if (centroid)
    {
      hipLaunchKernelGGL(k_add_flags, dim3(kmax, y), dim3(256), 0, st, b);
      hipLaunchKernelGGL(k_add_vscan, dim3(1, y), dim3(1024), 0, st, b);
    }
    hipLaunchKernelGGL(k_add_fold, dim3(kmax, y), dim3(256), 0, st, b);
    hipLaunchKernelGGL(k_add_merge, dim3(omax + kmax, y), dim3(256), lds, st, b);
    hipLaunchKernelGGL(k_add_commit, dim3(1, y), dim3(64), 0, st, b);
  }
  for (int i = 0; i < count; ++i)
  {
    gs[i]->cur = 1 - gs[i]->cur;
    gs[i]->n_upper += ns[i];
    const int rc = refresh_state(gs[i]);  // whether a point changed (the kd-tree is only dropped then, :315-317) is read by lsa_device_grid_submap_valid
    if (rc) return rc;
  }
  return LSA_OK;
}
int add_batch(lsa_device_grid* g, int n, bool fixed, double time, bool do_roll) { return add_batches(&g, &n, 1, fixed, time, do_roll); }

}  // namespace

extern "C" {

int lsa_device_grid_create(lsa_ctx* ctx, lsa_device_grid** out)
{
  if (!ctx || !out) return LSA_E_ARG;
  *out = nullptr;
  if (hipSetDevice(ctx->device) != hipSuccess) return LSA_E_HIP;
  lsa_device_grid* g = new lsa_device_grid;
  g->ctx = ctx;
  bool ok = hipMalloc((void**)&g->st, kStInts * sizeof(int)) == hipSuccess;
  ok = ok && hipHostMalloc((void**)&g->host_st, kStInts * sizeof(int), hipHostMallocDefault) == hipSuccess;
  ok = ok && hipHostMalloc((void**)&g->host_sub, sizeof(u64), hipHostMallocCoherent | hipHostMallocMapped) == hipSuccess;
  if (ok) *g->host_sub = 0;
  ok = ok && hipHostMalloc((void**)&g->host_ahead, 2 * sizeof(u64), hipHostMallocCoherent | hipHostMallocMapped) == hipSuccess;
  if (ok) g->host_ahead[0] = g->host_ahead[1] = 0;
  {
    // The maps' kernels go on the context's LOOK-AHEAD stream (next frame's extraction, next ego-motion targets): a
    // process has four hardware queues, and the registration's stream, the look-ahead stream and the copy stream are
    // busy beside the insertions -- every further stream shares a queue with one of them, and when that one is the
    // ICP's the frame pays (one box, alternating runs: ego-motion phase 0.42 ms per frame with the maps on the
    // look-ahead stream, 0.60 with a stream per map, 0.83 with one new stream for all maps; 890 / 765 / 665 frames/s).
    // The order on that stream is the order of need: insertions (end of frame f), extraction of frame f + 2 (announced
    // during frame f + 1), targets of frame f + 2.
    // LSA_MAP_STREAM=own|shared: a stream per map / one more stream for all maps (the experiments above)
    const char* e = std::getenv("LSA_MAP_STREAM");
    const std::string mode = e ? e : "prefetch";
    int least = 0, greatest = 0;
    (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
    if (mode == "own") { ok = ok && hipStreamCreateWithPriority(&g->stream, hipStreamNonBlocking, least) == hipSuccess; g->own_stream = true; }
    else if (mode != "shared" && ctx->prefetch_stream) g->stream = ctx->prefetch_stream;
    else
    {
      if (!ctx->map_stream) ok = ok && hipStreamCreateWithPriority(&ctx->map_stream, hipStreamNonBlocking, least) == hipSuccess;
      g->stream = ctx->map_stream;
      if (ok) { ctx->map_stream_users++; g->shared_stream = true; }
    }
  }
  for (hipEvent_t* e : {&g->ev_state, &g->ev_in, &g->ev_out, &g->ev_sub, &g->ev_ahead}) ok = ok && hipEventCreateWithFlags(e, hipEventDisableTiming) == hipSuccess;
  if (!ok) { lsa_device_grid_destroy(g); return LSA_E_HIP; }
  *out = g;
  return lsa_device_grid_reset(g, nullptr);
}

void lsa_device_grid_destroy(lsa_device_grid* g)
{
  if (!g) return;
  (void)hipSetDevice(g->ctx->device);
  if (g->stream) (void)hipStreamSynchronize(g->stream);
  (void)hipStreamSynchronize(g->ctx->stream);  // a match may still read a sub-map: nothing of the grid is in use after this
  free_view(g->buf[0]); free_view(g->buf[1]); free_view(g->fresh);
  auto fr = [](void* p) { if (p) (void)hipFree(p); };
  fr(g->st); fr(g->batch); fr(g->bkeys); fr(g->skeys); fr(g->border); fr(g->sorder); fr(g->heads); fr(g->fresh_flag); fr(g->vrank); fr(g->chunks); fr(g->old_local);
  if (g->host_st) (void)hipHostFree(g->host_st);
  if (g->host_sub) (void)hipHostFree(g->host_sub);
  if (g->host_ahead) (void)hipHostFree(g->host_ahead);
  for (hipEvent_t e : {g->ev_state, g->ev_in, g->ev_out, g->ev_sub, g->ev_ahead})
    if (e) (void)hipEventDestroy(e);
  if (g->stream && g->own_stream) (void)hipStreamDestroy(g->stream);
  if (g->shared_stream && --g->ctx->map_stream_users == 0 && g->ctx->map_stream)
  {
    (void)hipStreamDestroy(g->ctx->map_stream);
    g->ctx->map_stream = nullptr;
  }
  delete g;
}

// RollingGrid::Reset (RollingGrid.cxx:40-48): the map is emptied, the grid is centred on `position` (snapped to the
// voxel resolution)
int lsa_device_grid_reset(lsa_device_grid* g, const float position[3])
{
  if (!g) return LSA_E_ARG;
  G_HIP(hipSetDevice(g->ctx->device));
  int h[kStInts] = {0};
  const float r = (float)g->VoxelResolution;
  for (int d = 0; d < 3; ++d)
  {
    const float v = std::floor((position ? position[d] : 0.f) / r) * r;
    std::memcpy(&h[kStPosX + d], &v, sizeof(float));
    h[kStTmp + d] = 0x7fffffff;
    h[kStTmp + 3 + d] = (int)0x80000000;
  }
  G_HIP(hipStreamSynchronize(g->stream));
  G_HIP(hipMemcpy(g->st, h, sizeof(h), hipMemcpyHostToDevice));
  std::memcpy(g->host_st, h, sizeof(h));
  g->n_upper = 0;
  g->submap_valid = false;
  return LSA_OK;
}

int lsa_device_grid_clear(lsa_device_grid* g)
{
  if (!g) return LSA_E_ARG;
  G_HIP(hipSetDevice(g->ctx->device));
  {
    const int rc = after_submap(g);
    if (rc) return rc;
  }
  hipLaunchKernelGGL(k_set_int, dim3(1), dim3(64), 0, g->stream, g->st + kStN, 0);
  hipLaunchKernelGGL(k_set_int, dim3(1), dim3(64), 0, g->stream, g->st + kStNbPoints, 0);
  g->n_upper = 0;
  g->submap_valid = false;
  return refresh_state(g);
}

static int readd_everything(lsa_device_grid* g);

int lsa_device_grid_set(lsa_device_grid* g, const char* name, double value)
{
  if (!g || !name) return LSA_E_ARG;
  const std::string n(name);
  if (n == "LeafSize") { g->LeafSize = value; return LSA_OK; }
  if (n == "MinFramesPerVoxel") { g->MinFramesPerVoxel = (unsigned)value; return LSA_OK; }
  if (n == "Sampling") { g->Sampling = (int)value; return LSA_OK; }
  if (n == "DecayingThreshold") { g->DecayingThreshold = value; return LSA_OK; }
  if (n == "GridSize")
  {
    // RollingGrid::SetGridSize (:59-70): the points are put back into the resized grid
    g->GridSize = (int)value;
    return readd_everything(g);
  }
  if (n == "VoxelResolution")
  {
    // RollingGrid::SetVoxelResolution (:73-88): a multiple of the leaf size; the grid position is snapped to it
    g->VoxelResolution = int(value / g->LeafSize) * g->LeafSize;
    G_HIP(hipSetDevice(g->ctx->device));
    G_HIP(hipStreamSynchronize(g->stream));
    int h[kStInts];
    G_HIP(hipMemcpy(h, g->st, sizeof(h), hipMemcpyDeviceToHost));
    const float r = (float)g->VoxelResolution;
    for (int d = 0; d < 3; ++d)
    {
      float v;
      std::memcpy(&v, &h[kStPosX + d], sizeof(float));
      v = std::floor(v / r) * r;
      std::memcpy(&h[kStPosX + d], &v, sizeof(float));
    }
    G_HIP(hipMemcpy(g->st, h, sizeof(h), hipMemcpyHostToDevice));
    return readd_everything(g);
  }
  return g->ctx->fail(LSA_E_ARG, "lsa_device_grid_set: unknown parameter " + n);
}

double lsa_device_grid_get_param(const lsa_device_grid* g, const char* name)
{
  if (!g || !name) return 0.;
  const std::string n(name);
  if (n == "LeafSize") return g->LeafSize;
  if (n == "MinFramesPerVoxel") return g->MinFramesPerVoxel;
  if (n == "Sampling") return g->Sampling;
  if (n == "DecayingThreshold") return g->DecayingThreshold;
  if (n == "GridSize") return g->GridSize;
  if (n == "VoxelResolution") return g->VoxelResolution;
  return 0.;
}

// RollingGrid::Size() as of the last modification that has completed on the device (waits for it)
int lsa_device_grid_size(lsa_device_grid* g)
{
  if (!g) return LSA_E_ARG;
  if (hipSetDevice(g->ctx->device) != hipSuccess || hipEventSynchronize(g->ev_state) != hipSuccess) return LSA_E_HIP;
  // every modification has landed: the bound on the number of voxels is the number itself.  (tighten() alone seldom found the
  // event complete in the pipeline -- something had always just been enqueued behind it --, the bound then grew by every
  // keyframe's points and the map's buffers were "outgrown" and doubled every twenty keyframes: a stall of the maps' stream,
  // three device-wide copies and eight buffers to free at the next frame's start, 0.3-0.6 ms each time.)
  if (g->n_upper > 0) g->n_upper = std::min(g->n_upper, std::max(g->host_st[kStN], 0));
  return g->host_st[kStNbPoints];
}


int lsa_device_grid_add(lsa_device_grid* g, const lsa_point_t* pts, int n, int fixed, double time, int roll_first)
{
  if (!g || n < 0 || (!pts && n > 0)) return g ? g->ctx->fail(LSA_E_ARG, "lsa_device_grid_add: bad argument") : LSA_E_ARG;
  if (n == 0) return LSA_OK;  // "Pointcloud is empty, voxel grid not updated."
  G_HIP(hipSetDevice(g->ctx->device));
  int rc = ensure_batch(g, n);
  if (rc) return rc;
  G_HIP(hipMemcpyAsync(g->batch, pts, (size_t)n * sizeof(lsa_point_t), hipMemcpyHostToDevice, g->stream));
  G_HIP(hipStreamSynchronize(g->stream));  // pts may be pageable and go away
  return add_batch(g, n, fixed != 0, time, roll_first != 0);
}

// the keypoints of a device set, moved by `pose` (WORLD), added without leaving the device: Slam::UpdateMapsUsingTworld
// (slam_lib/src/Slam.cxx:1178-1222).  In two steps for callers that hand the insertion to another host thread: _stage
// reads the context's keypoints (ordered behind what the context's stream has enqueued, and the context's stream behind
// it: the set may be rewritten right after), _add_staged is the insertion proper, on the grid's stream alone.
int lsa_device_grid_stage_keypoints(lsa_device_grid* g, int set, int type, const double pose[16])
{
  if (!g || !pose || set < 0 || set > 2 || type < 0 || type > 2) return g ? g->ctx->fail(LSA_E_ARG, "lsa_device_grid_stage_keypoints: bad argument") : LSA_E_ARG;
  lsa_ctx* ctx = g->ctx;
  const int n = ctx->kp_n[set][type];
  g->staged = 0;
  if (n <= 0) return LSA_OK;
  G_HIP(hipSetDevice(ctx->device));
  int rc = ensure_batch(g, n);
  if (rc) return rc;
  // the transform runs on the context's stream, in order with whatever rewrites the keypoints next (a few microseconds
  // on a stream that is idle at the end of a frame); the grid's stream only waits for it.  The batch buffer is free: the
  // last insertion that read it is over (ev_out, recorded behind every insertion).
  G_HIP(hipStreamWaitEvent(ctx->stream, g->ev_out, 0));
  rc = transform_points_to(ctx, ctx->kp[set][type], n, pose, reinterpret_cast<lsa_point_t*>(g->batch), ctx->stream);
  if (rc) return rc;
  g->staged = n;
  return order_after_context(g);
}
// ... of the keypoint types of a keyframe together: ONE transform launch for all the maps (a block row each), one event
int lsa_device_grid_stage_keypoints_all(lsa_device_grid* const* grids, const int* types, int count, int set, const double pose[16])
{
  if (!grids || !types || !pose || count < 1 || count > 3 || set < 0 || set > 2) return LSA_E_ARG;
  for (int i = 0; i < count; ++i)
    if (!grids[i] || types[i] < 0 || types[i] > 2 || grids[i]->ctx != grids[0]->ctx) return LSA_E_ARG;
  lsa_device_grid* g = grids[0];
  lsa_ctx* ctx = g->ctx;
  G_HIP(hipSetDevice(ctx->device));
  const lsa_point_t* src[3] = {nullptr, nullptr, nullptr};
  lsa_point_t* dst[3] = {nullptr, nullptr, nullptr};
  int ns[3] = {0, 0, 0};
  bool any = false;
  for (int i = 0; i < count; ++i)
  {
    lsa_device_grid* gi = grids[i];
    const int n = ctx->kp_n[set][types[i]];
    gi->staged = 0;
    if (n <= 0) continue;
    const int rc = ensure_batch(gi, n);
    if (rc) return rc;
    // (the batch buffer is free once the last insertion that read it is over: ev_out)
    G_HIP(hipStreamWaitEvent(ctx->stream, gi->ev_out, 0));
    src[i] = ctx->kp[set][types[i]];
    dst[i] = reinterpret_cast<lsa_point_t*>(gi->batch);
    ns[i] = n;
    any = true;
  }
  if (!any) return LSA_OK;
  const int rc = transform_sets_to(ctx, src, ns, pose, dst, ctx->stream);
  if (rc) return rc;
  G_HIP(hipEventRecord(g->ev_in, ctx->stream));
  for (int i = 0; i < count; ++i)
  {
    grids[i]->staged = ns[i];
    if (ns[i] > 0) G_HIP(hipStreamWaitEvent(grids[i]->stream, g->ev_in, 0));
  }
  return LSA_OK;
}
int lsa_device_grid_add_staged(lsa_device_grid* g, double time)
{
  if (!g) return LSA_E_ARG;
  const int n = g->staged;
  g->staged = 0;
  if (n <= 0) return LSA_OK;  // "Pointcloud is empty, voxel grid not updated."
  G_HIP(hipSetDevice(g->ctx->device));
  return add_batch(g, n, false, time, true);
}
// ... of several maps of one context at once (the keypoint types of a keyframe): one launch of every step for all of them
int lsa_device_grid_add_staged_all(lsa_device_grid* const* grids, int count, double time)
{
  if (!grids || count < 1 || count > 3) return LSA_E_ARG;
  lsa_device_grid* gs[3];
  int ns[3], m = 0;
  for (int i = 0; i < count; ++i)
  {
    if (!grids[i]) return LSA_E_ARG;
    if (grids[i]->staged > 0) { gs[m] = grids[i]; ns[m] = grids[i]->staged; ++m; }  // "Pointcloud is empty, voxel grid not updated."
    grids[i]->staged = 0;
  }
  if (m == 0) return LSA_OK;
  if (hipSetDevice(gs[0]->ctx->device) != hipSuccess) return LSA_E_HIP;
  return add_batches(gs, ns, m, false, time, true);
}
int lsa_device_grid_add_keypoints(lsa_device_grid* g, int set, int type, const double pose[16], double time)
{
  const int rc = lsa_device_grid_stage_keypoints(g, set, type, pose);
  return rc ? rc : lsa_device_grid_add_staged(g, time);
}

int lsa_device_grid_roll(lsa_device_grid* g, const float mn[3], const float mx[3])
{
  if (!g || !mn || !mx) return LSA_E_ARG;
  G_HIP(hipSetDevice(g->ctx->device));
  int rc = after_submap(g);
  if (rc) return rc;
  rc = ensure_map(g, std::max(g->n_upper, 1));
  if (rc) return rc;
  int box[6];
  for (int d = 0; d < 3; ++d)
  {
    int a, b;
    std::memcpy(&a, &mn[d], sizeof(int));
    std::memcpy(&b, &mx[d], sizeof(int));
    box[d] = a >= 0 ? a : a ^ 0x7fffffff;
    box[3 + d] = b >= 0 ? b : b ^ 0x7fffffff;
  }
  G_HIP(hipMemcpyAsync(g->st + kStTmp, box, sizeof(box), hipMemcpyHostToDevice, g->stream));
  G_HIP(hipStreamSynchronize(g->stream));
  rc = roll(g, true);
  if (rc) return rc;
  return refresh_state(g);
}

int lsa_device_grid_clear_old_points(lsa_device_grid* g, double now)
{
  if (!g) return LSA_E_ARG;
  G_HIP(hipSetDevice(g->ctx->device));
  int rc = after_submap(g);
  if (rc) return rc;
  rc = ensure_map(g, std::max(g->n_upper, 1));
  if (rc) return rc;
  const MapView src = g->buf[g->cur], dst = g->buf[1 - g->cur];
  compact(g, DecayPred{src.pts, now, g->DecayingThreshold}, CopyEmit{src, dst}, g->st + kStN, std::max(g->n_upper, 1), g->st + kStN);
  g->cur = 1 - g->cur;
  return refresh_state(g);
}

// RollingGrid::Get(clean) (:95-114) in key order; returns the number of points written
int lsa_device_grid_get(lsa_device_grid* g, int clean, lsa_point_t* out, int capacity)
{
  if (!g || (!out && capacity > 0)) return LSA_E_ARG;
  lsa_ctx* ctx = g->ctx;
  G_HIP(hipSetDevice(ctx->device));
  if (g->n_upper == 0) return 0;
  int rc = ensure_map(g, g->n_upper);
  if (rc) return rc;
  rc = ensure_scratch(ctx, (size_t)g->n_upper * sizeof(lsa_point_t));
  if (rc) return rc;
  const MapView m = g->buf[g->cur];
  rc = order_after_context(g);  // the scratch buffer is the context's; a sub-map extraction on its stream comes first too
  if (rc) return rc;
  SubMapPred pred{m.keys, m.pts, m.count, g->st, BoxArg{}, nullptr, nullptr, g->GridSize, (float)g->VoxelResolution, g->VoxelResolution, 0, g->MinFramesPerVoxel, -1, clean ? 3 : 0};
  compact(g, pred, PointEmit{m.pts, reinterpret_cast<float4*>(ctx->scratch_out)}, g->st + kStN, g->n_upper, g->st + kStSub);
  int kept = 0;
  G_HIP(hipMemcpyAsync(&kept, g->st + kStSub, sizeof(int), hipMemcpyDeviceToHost, g->stream));
  G_HIP(hipStreamSynchronize(g->stream));
  const int n = std::min(kept, capacity);
  if (n > 0) G_HIP(hipMemcpy(out, ctx->scratch_out, (size_t)n * sizeof(lsa_point_t), hipMemcpyDeviceToHost));
  return n;
}

// RollingGrid::BuildSubMapKdTree (:353-442): the sub-map becomes the kNN target (slot, type) of the context without
// leaving the device -- the points in key order, the search grid is built with the next match.  _begin enqueues it (on
// the grid's stream; the context's stream goes on behind it), _end waits for its size: several grids build side by
// side and are waited for once.  The box: mn/mx given; or, box_type >= 0, the box of that keypoint type as
// lsa_keypoint_bboxes_begin left it on the device (nothing is read back); or none: the whole map.
static int build_submap_begin(lsa_device_grid* g, const float mn[3], const float mx[3], int box_type, int min_nb_points, int slot, int type)
{
  if (!g || slot < 0 || slot > 1 || type < 0 || type > 2 || (mn && !mx) || box_type > 2)
    return g ? g->ctx->fail(LSA_E_ARG, "lsa_device_grid_build_submap: bad argument") : LSA_E_ARG;
  lsa_ctx* ctx = g->ctx;
  if (g->sub_target >= 0) return ctx->fail(LSA_E_STATE, "lsa_device_grid_build_submap_begin: the previous one has not been ended");
  G_HIP(hipSetDevice(ctx->device));
  const int ti = slot * 3 + type;
  g->sub_target = ti;
  g->submap_valid = true;
  tighten(g);
  if (g->n_upper == 0) return LSA_OK;
  int rc = ensure_map(g, g->n_upper);
  if (rc) return rc;
  rc = ensure_target(ctx, ti, g->n_upper);
  if (rc) return rc;
  // The extraction runs on the CONTEXT's stream, behind the grid's last modification (ev_out): the box words, the target
  // and the next match are the context's anyway, so nothing else has to be ordered, and the size comes back through
  // coherent host memory -- no copy, no event, no host call between the kernels.
  hipStream_t st = ctx->stream;
  G_HIP(hipStreamWaitEvent(st, g->ev_out, 0));
  G_HIP(hipStreamWaitEvent(st, g->ev_ahead, 0));  // an extraction ahead of time that was not taken over shares the scratch
  g->ahead_phase = 0;
  Target& t = ctx->target[ti];
  const MapView m = g->buf[g->cur];
  const bool boxed = mn || box_type >= 0;
  SubMapPred pred{m.keys, m.pts, m.count, g->st, BoxArg{}, nullptr, nullptr, g->GridSize, (float)g->VoxelResolution, g->VoxelResolution, 0, g->MinFramesPerVoxel, min_nb_points, boxed ? 1 : 0};
  bool filtered = false;
  if (boxed)
  {
    if (mn) for (int d = 0; d < 3; ++d) { pred.box.mn[d] = mn[d]; pred.box.mx[d] = mx[d]; }
    pred.ctx_box = mn ? nullptr : lsa::current_box_words(ctx) + 6 * box_type;
    filtered = !(min_nb_points < 0 || g->MinFramesPerVoxel <= 1);
    pred.mode = filtered ? 1 : 0;
  }
  const unsigned tag = ++g->sub_tag;
  {
    ProfScope ps(ctx, "map_submap", (double)g->n_upper * 44, st);
    // the sub-map is of the map as it is now: the changes the Adds before it flagged are in it (the flag goes with the
    // last kernel)
    compact(g, pred, PointEmit{m.pts, reinterpret_cast<float4*>(t.pts)}, g->st + kStN, g->n_upper, g->st + kStSub, false, true, st, filtered ? nullptr : g->host_sub,
            tag, filtered ? nullptr : g->st + kStUpdated);
    if (filtered)
    {
      // "Moving objects constraint was too strong, removing constraint": the rejected voxels follow when too few stayed
      pred.mode = 2;
      // the second pass appends behind what the first one kept (its predicate reads the first pass's count from a slot
      // of its own: the total moves while it runs)
      hipLaunchKernelGGL(k_copy_int, dim3(1), dim3(64), 0, st, g->st + kStSubFirst, g->st + kStSub);
      compact(g, pred, PointEmit{m.pts, reinterpret_cast<float4*>(t.pts)}, g->st + kStN, g->n_upper, g->st + kStSub, true, true, st, g->host_sub, tag, g->st + kStUpdated);
    }
  }
  G_HIP(hipEventRecord(g->ev_sub, st));  // the grid's next modification comes behind the extraction
  g->sub_pending = true;
  return LSA_OK;
}
int lsa_device_grid_build_submap_begin(lsa_device_grid* g, const float mn[3], const float mx[3], int min_nb_points, int slot, int type)
{
  return build_submap_begin(g, mn, mx, -1, min_nb_points, slot, type);
}
int lsa_device_grid_build_submap_begin_for_keypoints(lsa_device_grid* g, int box_type, int min_nb_points, int slot, int type)
{
  if (box_type < 0) return g ? g->ctx->fail(LSA_E_ARG, "lsa_device_grid_build_submap_begin_for_keypoints: bad argument") : LSA_E_ARG;
  if (g) g->ctx->bbox_pending = false;  // the box stays on the device: no lsa_keypoint_bboxes_end follows
  return build_submap_begin(g, nullptr, nullptr, box_type, min_nb_points, slot, type);
}
int lsa_device_grid_build_submap_end(lsa_device_grid* g)
{
  if (!g) return LSA_E_ARG;
  lsa_ctx* ctx = g->ctx;
  if (g->sub_target < 0) return ctx->fail(LSA_E_STATE, "lsa_device_grid_build_submap_end: no lsa_device_grid_build_submap_begin before");
  G_HIP(hipSetDevice(ctx->device));
  Target& t = ctx->target[g->sub_target];
  g->sub_target = -1;
  int kept = 0;
  if (g->sub_pending)
  {
    g->sub_pending = false;
    // {tag, size} arrives as one 8-byte store (bounded wait: 2 s)
    const auto t0 = std::chrono::steady_clock::now();
    unsigned spins = 0;
    while (true)
    {
      const u64 v = __atomic_load_n(g->host_sub, __ATOMIC_ACQUIRE);
      if ((unsigned)(v >> 32) == g->sub_tag) { kept = (int)(unsigned)(v & 0xffffffffull); break; }
      if ((++spins & 1023u) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(2))
      {
        G_HIP(hipStreamSynchronize(ctx->stream));  // surfaces a failed launch as an error rather than a timeout
        return ctx->fail(LSA_E_HIP, "lsa_device_grid_build_submap_end: the sub-map's size did not arrive");
      }
    }
    // every refresh of the state enqueued before the extraction has landed (the extraction came behind ev_out)
    G_HIP(hipEventSynchronize(g->ev_state));
  }
  else G_HIP(hipEventSynchronize(g->ev_state));
  t.m = kept;
  t.dirty = kept > 0;
  g->submap_count = kept;
  g->host_st[kStUpdated] = 0;
  return kept;
}
int lsa_device_grid_build_submap(lsa_device_grid* g, const float mn[3], const float mx[3], int min_nb_points, int slot, int type)
{
  const int rc = lsa_device_grid_build_submap_begin(g, mn, mx, min_nb_points, slot, type);
  return rc ? rc : lsa_device_grid_build_submap_end(g);
}

// ---- sub-maps ahead of time -------------------------------------------------------------------------------------------
// The sub-map the next localization will ask for only depends on the outer voxels its keypoints' box touches, and that box
// is known to a voxel long before the localization: _ahead_begin extracts the sub-map for the box of keypoint type
// `box_type` as lsa_keypoint_bboxes_begin(_interp) just left it on the device (the PREDICTED pose) into the context's
// spare map target, on the grid's stream behind the last insertion; _ahead_poll (non-blocking, call it now and then)
// enqueues the spare target's search grid once the extraction's size has arrived; _ahead_take, after
// lsa_keypoint_bboxes_begin under the ACTUAL pose, compares the two voxel ranges on the device and, when they are the
// same, swaps the spare target in as target (slot, type): *taken = 1, the return value is the sub-map's size, and
// lsa_device_grid_build_submap_begin / _end are not needed.  Anything that does not fit (*taken = 0) leaves everything
// as it was.  Same sub-map, byte for byte, either way.
int lsa_device_grid_submap_ahead_begin(lsa_device_grid* g, int box_type, int min_nb_points, int type)
{
  if (!g || box_type < 0 || box_type > 2 || type < 0 || type > 2) return g ? g->ctx->fail(LSA_E_ARG, "lsa_device_grid_submap_ahead_begin: bad argument") : LSA_E_ARG;
  lsa_ctx* ctx = g->ctx;
  G_HIP(hipSetDevice(ctx->device));
  g->ahead_phase = 0;
  ctx->bbox_pending = false;  // the box stays on the device
  tighten(g);
  if (g->n_upper == 0 || g->sub_target >= 0) return LSA_OK;
  int rc = ensure_map(g, g->n_upper);
  if (rc) return rc;
  if (ctx->map_ahead_ready[type]) { G_HIP(hipEventSynchronize(ctx->ev_map_ahead[type])); ctx->map_ahead_ready[type] = false; }
  rc = ensure_target(ctx, 9 + type, g->n_upper);
  if (rc) return rc;
  rc = after_submap(g);
  if (rc) return rc;
  // The box words: enqueued on this very stream by lsa_keypoint_boxes_predicted, or on the context's by
  // lsa_keypoint_bboxes_begin -- then this stream comes behind the context's.  (The spare target's last readers, searches of an
  // earlier frame, have long finished: every frame ends with the host reading its last solve's result.)
  if (!(ctx->pred_on_lookahead && g->stream == ctx->prefetch_stream))
  {
    rc = order_after_context(g);
    if (rc) return rc;
  }
  hipStream_t st = g->stream;
  // The predicted box becomes a range of outer voxels on the grid's stream, behind the last insertion (which may move the
  // grid).  The words are rewritten for the actual box later: should this kernel be so late that it reads those, or a
  // half-written box, the extraction below is simply for the range it stored, and _ahead_take compares the actual range
  // with the stored one -- a wrong guess costs the extraction, never the result.
  const unsigned* words = reinterpret_cast<const unsigned*>(ctx->range_bits + 16) + 6 * box_type;
  hipLaunchKernelGGL(k_pred_box, dim3(1), dim3(64), 0, st, words, g->GridSize, (float)g->VoxelResolution, g->VoxelResolution, g->st);
  Target& t = ctx->target[9 + type];
  const MapView m = g->buf[g->cur];
  const bool filtered = !(min_nb_points < 0 || g->MinFramesPerVoxel <= 1);
  SubMapPred pred{m.keys, m.pts, m.count, g->st, BoxArg{}, nullptr, g->st + kStPred, g->GridSize, (float)g->VoxelResolution, g->VoxelResolution, filtered ? 1 : 0,
                  g->MinFramesPerVoxel, min_nb_points, 1};
  const unsigned tag = ++g->ahead_tag;
  {
    ProfScope ps(ctx, "map_submap_ahead", (double)g->n_upper * 44, st);
    compact(g, pred, PointEmit{m.pts, reinterpret_cast<float4*>(t.pts)}, g->st + kStN, g->n_upper, g->st + kStSub, false, true, st, filtered ? nullptr : g->host_ahead, tag);
    if (filtered)
    {
      pred.mode = 2;
      hipLaunchKernelGGL(k_copy_int, dim3(1), dim3(64), 0, st, g->st + kStSubFirst, g->st + kStSub);
      compact(g, pred, PointEmit{m.pts, reinterpret_cast<float4*>(t.pts)}, g->st + kStN, g->n_upper, g->st + kStSub, true, true, st, g->host_ahead, tag);
    }
  }
  G_HIP(hipEventRecord(g->ev_ahead, st));  // whoever uses the grid's scratch next on another stream comes behind this
  g->ahead_phase = 1;
  g->ahead_type = type;
  g->ahead_min = min_nb_points;
  return LSA_OK;
}
int lsa_device_grid_submap_ahead_poll(lsa_device_grid* g)
{
  if (!g) return LSA_E_ARG;
  if (g->ahead_phase != 1) return g->ahead_phase;
  const u64 v = __atomic_load_n(g->host_ahead, __ATOMIC_ACQUIRE);
  if ((unsigned)(v >> 32) != g->ahead_tag) return 1;
  lsa_ctx* ctx = g->ctx;
  G_HIP(hipSetDevice(ctx->device));
  const int type = g->ahead_type;
  Target& t = ctx->target[9 + type];
  g->ahead_m = (int)(unsigned)(v & 0xffffffffull);
  t.m = g->ahead_m;
  t.cell_hint = ctx->target[LSA_TARGET_MAP * 3 + type].cell_hint;
  t.dirty = false;
  if (t.m > 0)
  {
    const int tis[1] = {9 + type};
    const int rc = build_target_grids(ctx, tis, 1, g->stream);
    if (rc) return rc;
  }
  G_HIP(hipEventRecord(ctx->ev_map_ahead[type], g->stream));
  g->ahead_phase = 2;
  return 2;
}
// The same for several maps of one context at once: once ALL their sizes have arrived their search grids are built by ONE
// sequence of launches (a block row per target) instead of one sequence each.  Returns 1 while a size is missing, 2 when the
// grids are enqueued (or nothing was pending).
int lsa_device_grid_submap_ahead_poll_all(lsa_device_grid* const* grids, int count)
{
  if (!grids || count < 1 || count > 3) return LSA_E_ARG;
  lsa_ctx* ctx = nullptr;
  hipStream_t st = nullptr;
  u64 v[3];
  bool pending[3] = {false, false, false}, any = false;
  for (int i = 0; i < count; ++i)
  {
    lsa_device_grid* g = grids[i];
    if (!g) return LSA_E_ARG;
    if (g->ahead_phase != 1) continue;
    if (ctx && (g->ctx != ctx || g->stream != st)) return g->ctx->fail(LSA_E_ARG, "lsa_device_grid_submap_ahead_poll_all: maps of different contexts or streams");
    ctx = g->ctx;
    st = g->stream;
    v[i] = __atomic_load_n(g->host_ahead, __ATOMIC_ACQUIRE);
    if ((unsigned)(v[i] >> 32) != g->ahead_tag) return 1;
    pending[i] = any = true;
  }
  if (!any) return 2;
  LSA_HIP(ctx, hipSetDevice(ctx->device));
  int tis[3], nt = 0;
  for (int i = 0; i < count; ++i)
  {
    if (!pending[i]) continue;
    lsa_device_grid* g = grids[i];
    const int type = g->ahead_type;
    Target& t = ctx->target[9 + type];
    g->ahead_m = (int)(unsigned)(v[i] & 0xffffffffull);
    t.m = g->ahead_m;
    t.cell_hint = ctx->target[LSA_TARGET_MAP * 3 + type].cell_hint;
    t.dirty = false;
    if (t.m > 0) tis[nt++] = 9 + type;
  }
  if (nt > 0)
  {
    const int rc = build_target_grids(ctx, tis, nt, st);
    if (rc) return rc;
  }
  for (int i = 0; i < count; ++i)
  {
    if (!pending[i]) continue;
    LSA_HIP(ctx, hipEventRecord(ctx->ev_map_ahead[grids[i]->ahead_type], st));
    grids[i]->ahead_phase = 2;
  }
  return 2;
}
// ... or waited for (a thread that has nothing else to do): returns once the search grid has been enqueued
int lsa_device_grid_submap_ahead_wait(lsa_device_grid* g)
{
  if (!g) return LSA_E_ARG;
  const auto t0 = std::chrono::steady_clock::now();
  unsigned spins = 0;
  while (g->ahead_phase == 1)
  {
    const int rc = lsa_device_grid_submap_ahead_poll(g);
    if (rc < 0) return rc;
    if (rc == 1 && (++spins & 255u) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(500))
      return g->ctx->fail(LSA_E_HIP, "lsa_device_grid_submap_ahead_wait: the extraction's size did not arrive");
#if defined(__x86_64__)
    __builtin_ia32_pause();
#endif
  }
  return g->ahead_phase;
}
// _take in two steps, so that the comparisons of several maps are enqueued before any of them is waited for: _take_begin
// returns 1 when a comparison is on its way (0: nothing fits, extract the sub-map as usual), _take_end waits for it.
int lsa_device_grid_submap_ahead_take_begin(lsa_device_grid* g, int box_type, int min_nb_points, int slot, int type)
{
  if (!g || box_type < 0 || box_type > 2 || slot < 0 || slot > 1 || type < 0 || type > 2)
    return g ? g->ctx->fail(LSA_E_ARG, "lsa_device_grid_submap_ahead_take: bad argument") : LSA_E_ARG;
  lsa_ctx* ctx = g->ctx;
  g->take_pending = false;
  if (g->ahead_phase == 1)
  {
    const int rc = lsa_device_grid_submap_ahead_poll(g);
    if (rc < 0) return rc;
  }
  const bool fits = g->ahead_phase == 2 && g->ahead_type == type && g->ahead_min == min_nb_points && g->sub_target < 0 &&
                    ctx->target[9 + type].cell_hint == ctx->target[slot * 3 + type].cell_hint;
  g->ahead_phase = 0;
  if (!fits) return 0;
  G_HIP(hipSetDevice(ctx->device));
  // the comparison runs on the context's stream, where the actual box was just enqueued, behind the grid's stream (the
  // predicted range and the state it reads)
  G_HIP(hipStreamWaitEvent(ctx->stream, ctx->ev_map_ahead[type], 0));
  const unsigned tag = ++g->ahead_tag;
  const unsigned* words = lsa::current_box_words(ctx) + 6 * box_type;
  hipLaunchKernelGGL(k_box_check, dim3(1), dim3(64), 0, ctx->stream, words, g->GridSize, (float)g->VoxelResolution, g->VoxelResolution, g->st, g->host_ahead + 1, tag);
  g->take_pending = true;
  g->take_slot = slot;
  return 1;
}
int lsa_device_grid_submap_ahead_take_end(lsa_device_grid* g, int* taken)
{
  if (!g || !taken) return LSA_E_ARG;
  *taken = 0;
  if (!g->take_pending) return LSA_OK;
  g->take_pending = false;
  lsa_ctx* ctx = g->ctx;
  const int type = g->ahead_type, slot = g->take_slot;
  const unsigned tag = g->ahead_tag;
  const auto t0 = std::chrono::steady_clock::now();
  unsigned spins = 0;
  u64 v;
  while ((unsigned)((v = __atomic_load_n(g->host_ahead + 1, __ATOMIC_ACQUIRE)) >> 32) != tag)
    if ((++spins & 1023u) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(2))
    {
      G_HIP(hipStreamSynchronize(ctx->stream));
      return ctx->fail(LSA_E_HIP, "lsa_device_grid_submap_ahead_take: the comparison did not arrive");
    }
  if (!(v & 1ull)) return LSA_OK;  // another range of voxels: the caller extracts the sub-map now
  ctx->bbox_pending = false;
  std::swap(ctx->target[slot * 3 + type], ctx->target[9 + type]);
  ctx->target[slot * 3 + type].dirty = false;
  g->submap_valid = true;
  g->submap_count = g->ahead_m;
  G_HIP(hipEventSynchronize(g->ev_state));  // the flag's last refresh has landed: it was taken back by the comparison
  g->host_st[kStUpdated] = 0;
  G_HIP(hipEventRecord(g->ev_sub, ctx->stream));
  *taken = 1;
  return g->ahead_m;
}
int lsa_device_grid_submap_ahead_take(lsa_device_grid* g, int box_type, int min_nb_points, int slot, int type, int* taken)
{
  if (!taken) return LSA_E_ARG;
  *taken = 0;
  const int rc = lsa_device_grid_submap_ahead_take_begin(g, box_type, min_nb_points, slot, type);
  return rc <= 0 ? rc : lsa_device_grid_submap_ahead_take_end(g, taken);
}

// RollingGrid::IsSubMapKdTreeValid(): an Add that changed a voxel's point has dropped the sub-map (RollingGrid.cxx:315-317);
// rolling and decay do not (as in the reference).  Waits for the modifications enqueued so far.
int lsa_device_grid_submap_valid(lsa_device_grid* g)
{
  if (!g) return 0;
  if (hipSetDevice(g->ctx->device) != hipSuccess || hipEventSynchronize(g->ev_state) != hipSuccess) return 0;
  if (g->host_st[kStUpdated]) g->submap_valid = false;  // the flag is taken back by the next sub-map (lsa_device_grid_build_submap_begin)
  return g->submap_valid && g->submap_count > 0 ? 1 : 0;  // an empty sub-map counts as invalid (RollingGrid.h:154)
}

static int readd_everything(lsa_device_grid* g)
{
  // prevMap = Get(); Clear(); Add(prevMap)
  std::vector<lsa_point_t> all(std::max(g->n_upper, 1));
  const int n = lsa_device_grid_get(g, 0, all.data(), (int)all.size());
  if (n < 0) return n;
  int rc = lsa_device_grid_clear(g);
  if (rc) return rc;
  if (n > 0) return lsa_device_grid_add(g, all.data(), n, 0, -1., 1);
  return LSA_OK;
}

}  // extern "C"
