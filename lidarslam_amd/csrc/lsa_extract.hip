// lsa_extract.hip -- SpinningSensorKeypointExtractor::ComputeKeyPoints on the GPU
// (slam_lib/src/SpinningSensorKeypointExtractor.cxx:118-136 and everything it calls).
//
// Kernels (all HBM-bound gather / stencil / scan work, no MFMA):
//   k_ring_hist / k_ring_scan / k_ring_scatter   ConvertAndSortScanLines (SSKE.cxx:139-171):
//       stable counting sort of the firing-order AoS scan into ring-major float4 SoA
//   k_invalidate                                 InvalidateNotUsablePoints (SSKE.cxx:207-308)
//   k_curvature<MAXW>                            ComputeCurvature + LineFitting (SSKE.cxx:33-115, 311-471)
//   k_label                                      SetKeyPointsLabels (SSKE.cxx:474-573): the greedy, sorted
//       non-maximum selection is evaluated WITHOUT sorting, as the fixed point of a local rule in LDS
//       (a candidate wins once no undecided higher-priority candidate is left in its window)
//   k_compact                                    keypoint clouds, ring-major / index ascending (SSKE.cxx:575-589)
#include <cmath>
#include "lsa_ctx.h"
#include "lsa_device_math.h"

using namespace lsa;

namespace
{

#ifndef LSA_LABEL_THREADS
#define LSA_LABEL_THREADS 1024
#endif
constexpr int kLabelThreads = LSA_LABEL_THREADS;  // one block per ring

struct ExtractConst
{
  int W;
  float min_dist;
  float max_pos_diff_coeff;
  float line_max_sin;
  float line_sq_max_dist;
  float sq_dist_to_line_thr;
  float plane_thr;
  float edge_angle_thr;
  float sq_depth_gap_thr;
  float sq_saliency_thr;
  float intensity_thr;
};

__device__ __forceinline__ unsigned laser_of(const float4& second) { return __float_as_uint(second.w) & 0xffffu; }

// --------------------------------------------------------------------------------------------
// arms the read-back block of one extraction: counts 0, ring_meta 0 with max laser id -1, empty time range
__global__ void k_extract_init(int* __restrict__ out)
{
  const int i = threadIdx.x;
  if (i >= 16) return;
  int v = 0;
  if (i == 5) v = -1;                      // ring_meta[1]: max laser id
  if (i == 12 || i == 13) v = -1;          // time range lo = ~0ull
  out[i] = v;
}
// monotone encoding of doubles for 64-bit atomicMin/Max (as in lsa_transform.hip)
__device__ __forceinline__ unsigned long long time_bits(const float4& b)
{
  const unsigned long long u = (unsigned long long)__double_as_longlong(__hiloint2double(__float_as_int(b.y), __float_as_int(b.x)));
  return (u & 0x8000000000000000ull) ? ~u : (u | 0x8000000000000000ull);
}

__global__ __launch_bounds__(256) void k_ring_hist(const float4* __restrict__ frame, int n, uint32_t* __restrict__ block_hist,
                                                   int* __restrict__ ring_meta)
{
  __shared__ uint32_t h[kMaxRings];
  for (int r = threadIdx.x; r < kMaxRings; r += blockDim.x) h[r] = 0;
  __syncthreads();
  const int base = blockIdx.x * kBucketChunk;
  int mx = -1;
  for (int i = threadIdx.x; i < kBucketChunk; i += blockDim.x)
  {
    int g = base + i;
    if (g < n)
    {
      unsigned id = laser_of(frame[2 * (size_t)g + 1]);
      if (id < (unsigned)kMaxRings)
      {
        atomicAdd(&h[id], 1u);
        mx = max(mx, (int)id);
      }
      else
        atomicOr(&ring_meta[2], 1);
    }
  }
  for (int o = 32; o > 0; o >>= 1) mx = max(mx, __shfl_down(mx, o));
  if ((threadIdx.x & 63) == 0 && mx >= 0) atomicMax(&ring_meta[1], mx);
  __syncthreads();
  for (int r = threadIdx.x; r < kMaxRings; r += blockDim.x) block_hist[(size_t)blockIdx.x * kMaxRings + r] = h[r];
}

// One wavefront per ring: exclusive scan of the ring's column of the per-chunk histograms (where the ring's
// points of chunk b start inside the ring), the ring's length; ring 0 also publishes NbLaserRings.
__global__ __launch_bounds__(64) void k_ring_scan(uint32_t* __restrict__ block_hist, int nblocks, int* __restrict__ ring_len, int* __restrict__ ring_meta)
{
  const int r = blockIdx.x, lane = threadIdx.x;
  const int per = (nblocks + 63) / 64;
  const int b0 = min(nblocks, lane * per), b1 = min(nblocks, b0 + per);
  uint32_t mine = 0;
  for (int b = b0; b < b1; ++b) mine += block_hist[(size_t)b * kMaxRings + r];
  uint32_t inc = mine;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1)
  {
    const uint32_t t = __shfl_up(inc, o);
    if (lane >= o) inc += t;
  }
  uint32_t run = inc - mine;
  for (int b = b0; b < b1; ++b)
  {
    const uint32_t v = block_hist[(size_t)b * kMaxRings + r];
    block_hist[(size_t)b * kMaxRings + r] = run;
    run += v;
  }
  if (lane == 63) ring_len[r] = (int)inc;
  if (r == 0 && lane == 0) ring_meta[0] = ring_meta[1] + 1;  // NbLaserRings = max laser id + 1 (SSKE.cxx:153-164)
}

// lanes of the wavefront holding the same key (9-bit ring id): 9 ballots
__device__ __forceinline__ unsigned long long same_ring_lanes(unsigned id, bool active)
{
  unsigned long long m = __ballot(active);
#pragma unroll
  for (int bit = 0; bit < 9; ++bit)
  {
    const unsigned long long bal = __ballot(active && ((id >> bit) & 1u));
    m &= ((id >> bit) & 1u) ? bal : ~bal;
  }
  return m;
}

// Stable scatter of one chunk of the scan into ring-major order.  Wave 0 ranks the chunk 64 points at a time,
// in arrival order: a point's slot is ring_start + (points of its ring in earlier chunks) + (in earlier
// batches of this chunk, a running count in LDS) + (in lower lanes of its batch, a popcount).  Then the whole
// block moves the points.  Every block derives ring_start from ring_len itself (a 512-entry scan in LDS);
// block 0 publishes it for the later kernels.
__global__ __launch_bounds__(256) void k_ring_scatter(const float4* __restrict__ frame, int n, const uint32_t* __restrict__ block_hist,
                                                      const int* __restrict__ ring_len, int* __restrict__ ring_start,
                                                      float4* __restrict__ xyzi, uint32_t* __restrict__ orig, uint16_t* __restrict__ ring_of,
                                                      uint8_t* __restrict__ valid)
{
  __shared__ uint16_t ids[kBucketChunk];
  __shared__ uint32_t dest[kBucketChunk];
  __shared__ uint32_t slot[kMaxRings];  // next free slot of every ring for this chunk
  __shared__ uint32_t wcnt[4][kMaxRings];   // points of every ring in each wavefront's quarter of the chunk ...
  __shared__ uint32_t wslot[4][kMaxRings];  // ... and the next free slot of every ring for that quarter
  __shared__ int scan[kMaxRings];
  static_assert(kBucketChunk % 256 == 0, "a quarter of the chunk per wavefront, whole batches of 64");
  const int base = blockIdx.x * kBucketChunk;
  const int cnt = min(kBucketChunk, n - base);
  for (int i = threadIdx.x; i < cnt; i += blockDim.x) ids[i] = (uint16_t)laser_of(frame[2 * (size_t)(base + i) + 1]);
  for (int r = threadIdx.x; r < kMaxRings; r += blockDim.x) scan[r] = ring_len[r];
  __syncthreads();
  // inclusive scan of the 512 ring lengths, two per thread
  {
    const int t = threadIdx.x;
    const int a0 = scan[2 * t], a1 = scan[2 * t + 1];
    __shared__ int pair[256];
    pair[t] = a0 + a1;
    __syncthreads();
    for (int o = 1; o < 256; o <<= 1)
    {
      const int v = (t >= o) ? pair[t - o] : 0;
      __syncthreads();
      pair[t] += v;
      __syncthreads();
    }
    const int before = pair[t] - (a0 + a1);
    const uint32_t s0 = (uint32_t)before, s1 = (uint32_t)(before + a0);
    slot[2 * t] = s0 + block_hist[(size_t)blockIdx.x * kMaxRings + 2 * t];
    slot[2 * t + 1] = s1 + block_hist[(size_t)blockIdx.x * kMaxRings + 2 * t + 1];
    if (blockIdx.x == 0)
    {
      ring_start[2 * t] = (int)s0;
      ring_start[2 * t + 1] = (int)s1;
      if (t == 255) ring_start[kMaxRings] = pair[t];
    }
  }
  __syncthreads();
  // The four wavefronts rank a quarter of the chunk each (the first version left it all to the first one: sixteen batches
  // one after the other): the rings' points per quarter are counted first, a quarter's slots start behind the quarters
  // in front of it.
  constexpr int kQuarter = kBucketChunk / 4;
  {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int r = threadIdx.x; r < 4 * kMaxRings; r += blockDim.x) wcnt[r / kMaxRings][r % kMaxRings] = 0u;
    __syncthreads();
    for (int i = wave * kQuarter + lane; i < min(cnt, (wave + 1) * kQuarter); i += 64)
      if (ids[i] < kMaxRings) atomicAdd(&wcnt[wave][ids[i]], 1u);
    __syncthreads();
    for (int r = threadIdx.x; r < kMaxRings; r += blockDim.x)
    {
      uint32_t s = slot[r];
      for (int w = 0; w < 4; ++w) { wslot[w][r] = s; s += wcnt[w][r]; }
    }
    __syncthreads();
    const unsigned long long below = (1ull << lane) - 1ull;
    uint32_t* const myslot = wslot[wave];
    for (int b = wave * kQuarter; b < min(cnt, (wave + 1) * kQuarter); b += 64)
    {
      const int i = b + lane;
      const bool in = i < cnt && i < (wave + 1) * kQuarter;
      const unsigned id = in ? ids[i] : 0u;
      const bool active = in && id < (unsigned)kMaxRings;
      const unsigned long long same = same_ring_lanes(id, active);
      if (active)
      {
        const uint32_t s = myslot[id];
        dest[i] = s + (uint32_t)__popcll(same & below);
        // the last lane of the ring in this batch advances the ring's slot for the next batch
        if ((same >> lane) == 1ull) myslot[id] = s + (uint32_t)__popcll(same);
      }
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < cnt; i += blockDim.x)
  {
    if (ids[i] >= kMaxRings) continue;
    const float4 a = frame[2 * (size_t)(base + i)];
    const float4 b = frame[2 * (size_t)(base + i) + 1];
    const uint32_t d = dest[i];
    xyzi[d] = make_float4(a.x, a.y, a.z, b.z);
    orig[d] = (uint32_t)(base + i);
    ring_of[d] = ids[i];
    valid[d] = 7;
  }
}

// --------------------------------------------------------------------------------------------
__device__ __forceinline__ float sqdist3(const float4& a, const float4& b)
{
  Vec3<float> d = {a.x - b.x, a.y - b.y, a.z - b.z};
  return vsqnorm(d);
}
__device__ __forceinline__ float norm3(const float4& a)
{
  Vec3<float> d = {a.x, a.y, a.z};
  return vnorm(d);
}

__global__ __launch_bounds__(256) void k_invalidate(const float4* __restrict__ xyzi, const uint16_t* __restrict__ ring_of,
                                                    const int* __restrict__ ring_start, const int* __restrict__ ring_len, int n,
                                                    ExtractConst c, uint8_t* __restrict__ valid)
{
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  const int r = ring_of[j];
  const int idx = j - ring_start[r];
  const int np = ring_len[r];
  const int W = c.W;
  if (np < 2 * W + 1 || idx < W || idx >= np - W)
  {
    valid[j] = 0;
    return;
  }
  const float4 cur = xyzi[j];
  const float L = norm3(cur);
  if (L < c.min_dist) valid[j] = 0;
  const float lp = L * c.max_pos_diff_coeff;
  const float maxPosDiff = (lp < 0.02f) ? 0.02f : lp;  // std::max(L * coeff, 0.02f)
  const float sq = maxPosDiff * maxPosDiff;
  const float4 nxt = xyzi[j + 1];
  if (sqdist3(nxt, cur) > sq)
  {
    if (L < norm3(nxt))
    {
      valid[j + 1] = 0;
      for (int i = 1; i < W; ++i)
      {
        if (sqdist3(xyzi[j + i + 1], xyzi[j + i]) > sq) break;
        valid[j + i + 1] = 0;
      }
    }
    else
    {
      valid[j] = 0;
      for (int i = 1; i < W; ++i)
      {
        if (sqdist3(xyzi[j - i + 1], xyzi[j - i]) > sq) break;
        valid[j - i] = 0;
      }
    }
  }
}

// --------------------------------------------------------------------------------------------
struct Line
{
  Vec3<float> dir, pos;
};
__device__ __forceinline__ float line_sqdist(const Line& l, const Vec3<float>& p) { return vsqnorm(vcross(vsub(p, l.pos), l.dir)); }

// LineFitting::FitPCA on an accumulated covariance (SSKE.cxx:59-70)
__device__ __forceinline__ void line_from_cov(CovAccum<float>& acc, int cnt, Line& l)
{
  Sym3<float> cov;
  acc.finish(cnt, l.pos, cov);
  Vec3<float> e0, e1;
  float l0, l1, l2;
  eigen33<float>(cov, e0, e1, l.dir, l0, l1, l2);
}

// LineFitting::FitPCAAndCheckConsistency (SSKE.cxx:87-108) on the W neighbours of one side,
// p[0] nearest to the centre
template <int MAXW>
__device__ __forceinline__ bool fit_side(const Vec3<float> (&p)[MAXW], int W, const ExtractConst& c, Line& l)
{
  Vec3<float> last = p[0];
#pragma unroll
  for (int i = 1; i < MAXW; ++i)
    if (i == W - 1) last = p[i];
  const Vec3<float> U = normalized3(vsub(last, p[0]));
  bool straight = true;
#pragma unroll
  for (int i = 0; i + 1 < MAXW; ++i)
  {
    if (i + 1 < W && straight)
    {
      const Vec3<float> V = normalized3(vsub(p[i + 1], p[i]));
      const float sinAngle = vnorm(vcross(U, V));
      if (sinAngle > c.line_max_sin) straight = false;
    }
  }
  if (!straight) return false;
  CovAccum<float> acc;
#pragma unroll
  for (int i = 0; i < MAXW; ++i)
    if (i < W) acc.add(p[i].x, p[i].y, p[i].z);
  line_from_cov(acc, W, l);
  bool ok = true;
#pragma unroll
  for (int i = 0; i < MAXW; ++i)
    if (i < W && ok && line_sqdist(l, p[i]) > c.line_sq_max_dist) ok = false;
  return ok;
}

template <int MAXW>
__global__ __launch_bounds__(128) void k_curvature(const float4* __restrict__ xyzi, const uint16_t* __restrict__ ring_of,
                                                   const int* __restrict__ ring_start, const int* __restrict__ ring_len, int n,
                                                   ExtractConst c, const uint8_t* __restrict__ valid, float* __restrict__ o_angle,
                                                   float* __restrict__ o_gap, float* __restrict__ o_sal, float* __restrict__ o_int)
{
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  float angle = 0.f, gap = 0.f, sal = 0.f, igap = 0.f;
  const int r = ring_of[j];
  const int idx = j - ring_start[r];
  const int np = ring_len[r];
  const int W = c.W;
  if (!(np < 2 * W + 1) && idx >= W && idx + W < np && valid[j] != 0)
  {
    const float4 ctr4 = xyzi[j];
    const Vec3<float> ctr = {ctr4.x, ctr4.y, ctr4.z};
    Vec3<float> lp[MAXW], rp[MAXW];
    float i_prev = 0.f, i_next = 0.f;
#pragma unroll
    for (int i = 0; i < MAXW; ++i)
    {
      if (i < W)
      {
        const float4 a = xyzi[j - 1 - i];
        const float4 b = xyzi[j + 1 + i];
        lp[i] = {a.x, a.y, a.z};
        rp[i] = {b.x, b.y, b.z};
        if (i == 0) { i_prev = a.w; i_next = b.w; }
      }
      else
      {
        lp[i] = {0.f, 0.f, 0.f};
        rp[i] = {0.f, 0.f, 0.f};
      }
    }
    igap = fabsf(i_next - i_prev);

    Line ll, rl;
    const bool leftFlat = fit_side<MAXW>(lp, W, c, ll);
    const bool rightFlat = fit_side<MAXW>(rp, W, c, rl);
    float distLeft = 0.f, distRight = 0.f;
    if (leftFlat && rightFlat)
    {
      distLeft = line_sqdist(ll, ctr);
      distRight = line_sqdist(rl, ctr);
      if ((distLeft < c.sq_dist_to_line_thr) && (distRight < c.sq_dist_to_line_thr)) angle = vnorm(vcross(ll.dir, rl.dir));
    }
    else if (!leftFlat && rightFlat)
    {
      distLeft = 3.40282346638528859812e+38f;
#pragma unroll
      for (int i = 0; i < MAXW; ++i)
        if (i < W)
        {
          const float d = line_sqdist(rl, lp[i]);
          distLeft = (d < distLeft) ? d : distLeft;  // std::min(distLeft, d)
        }
      distLeft *= 0.25f;
    }
    else if (leftFlat && !rightFlat)
    {
      distRight = 3.40282346638528859812e+38f;
#pragma unroll
      for (int i = 0; i < MAXW; ++i)
        if (i < W)
        {
          const float d = line_sqdist(ll, rp[i]);
          distRight = (d < distRight) ? d : distRight;
        }
      distRight *= 0.25f;
    }
    else
    {
      // saliency: contiguous far neighbours (depth gap > 1.5 in squared norm), left then right,
      // accumulated straight into the covariance sums in the order the reference lists them
      const float sqCurrDepth = vsqnorm(ctr);
      CovAccum<float> acc;
      int cnt = 0;
      bool seen = false, stop = false;
#pragma unroll
      for (int i = 0; i < MAXW; ++i)
        if (i < W && !stop)
        {
          if (fabsf(vsqnorm(lp[i]) - sqCurrDepth) > 1.5f) { seen = true; acc.add(lp[i].x, lp[i].y, lp[i].z); ++cnt; }
          else if (seen) stop = true;
        }
      seen = false; stop = false;
#pragma unroll
      for (int i = 0; i < MAXW; ++i)
        if (i < W && !stop)
        {
          if (fabsf(vsqnorm(rp[i]) - sqCurrDepth) > 1.5f) { seen = true; acc.add(rp[i].x, rp[i].y, rp[i].z); ++cnt; }
          else if (seen) stop = true;
        }
      if (cnt > W)
      {
        Line fl;
        line_from_cov(acc, cnt, fl);
        sal = line_sqdist(fl, ctr);
      }
    }
    gap = (distLeft < distRight) ? distRight : distLeft;  // std::max(distLeft, distRight)
  }
  o_angle[j] = angle;
  o_gap[j] = gap;
  o_sal[j] = sal;
  o_int[j] = igap;
}

// --------------------------------------------------------------------------------------------
// Greedy non-maximum selection as a fixed point.  One 32-bit word per point holds state AND priority:
//   0 = out, kSel = selected, anything else = an undecided candidate whose priority is the word itself (larger wins).
// prio(k) > prio(j):  EDGE  score desc, index asc   (Utils::SortIdx(v, false) + stable tie-break)
//                     PLANE score asc,  index desc  (the same sorted array walked backwards)
// The word of a candidate is the order-preserving integer image of its score (inverted for planes); equal words are equal
// scores, and the index decides by the SIDE the neighbour is on.  Every thread owns PER consecutive points of the ring.
// Per round it reads its window (its points and hw on either side) ONCE into registers, sweeps its points forwards, then
// backwards, and writes the decisions it took: a decision is only ever taken on final facts (a selected neighbour; no
// undecided neighbour of higher priority), so a neighbour's state of the round's start gives correct, merely later,
// decisions.  (The first version kept state and score in two arrays and read every neighbour of every visit from LDS:
// 16 reads per point and sweep against 5 per thread and round.)
constexpr uint32_t kSel = 0xffffffffu;
constexpr int kNmsHalfMax = 8;  // NeighborWidth <= 8
template <bool PLANE>
__device__ __forceinline__ uint32_t nms_word(float v)
{
  const uint32_t u = __float_as_uint(v);
  const uint32_t o = u ^ ((u >> 31) ? 0xffffffffu : 0x80000000u);  // monotone in v; 0 and kSel only for NaN, which never qualifies
  return PLANE ? ~o : o;
}
// (not inlined: the four chunk sizes side by side in one function made the compiler spill the windows; called five times)
template <bool PLANE, int PER>
__device__ __noinline__ void nms_fixed_point(volatile uint32_t* w, int np, int hw)
{
  constexpr int WIN = PER + 2 * kNmsHalfMax;
  const int j0 = (int)threadIdx.x * PER;
#ifdef LSA_ABLATE_NMS_ROUNDS
  for (int round = 0; round < LSA_ABLATE_NMS_ROUNDS; ++round)  // (timing experiment only: wrong results)
#else
  while (true)
#endif
  {
    // a wavefront whose points are all decided has nothing to do but meet the others (decisions are final: it stays so);
    // the chains of undecided candidates are local, after two or three rounds most of the block's wavefronts are done
    bool mine = false;
#pragma unroll
    for (int a = 0; a < PER; ++a)
      if (j0 + a < np)
      {
        const uint32_t v = w[j0 + a];
        mine = mine || (v != 0u && v != kSel);
      }
    if (!__any(mine))
    {
      if (!__syncthreads_or(0)) break;
      continue;
    }
    uint32_t x[WIN];
#pragma unroll
    for (int i = 0; i < WIN; ++i)
    {
      const int k = j0 - kNmsHalfMax + i;
      x[i] = 0u;
      if (i >= kNmsHalfMax - hw && i < kNmsHalfMax + PER + hw && k >= 0 && k < np) x[i] = w[k];
    }
    int undecided = 0;
    auto visit = [&](int a) -> int {
      const int i = kNmsHalfMax + a;
      const uint32_t wj = x[i];
      if (wj == 0u || wj == kSel) return 0;
      // the largest word on either side says it all: kSel (the largest there is) = a selected neighbour; otherwise a word
      // above this one = an undecided neighbour of higher priority -- an equal score on the left (smaller index) wins among
      // edges, on the right among planes
      uint32_t ml = 0u, mr = 0u;
#pragma unroll
      for (int d = 1; d <= kNmsHalfMax; ++d)
        if (d <= hw)
        {
          ml = x[i - d] > ml ? x[i - d] : ml;
          mr = x[i + d] > mr ? x[i + d] : mr;
        }
      const bool selNear = ml == kSel || mr == kSel;
      const bool higher = PLANE ? (ml > wj || mr >= wj) : (ml >= wj || mr > wj);
      if (selNear) { x[i] = 0u; w[j0 + a] = 0u; return 0; }
      if (!higher) { x[i] = kSel; w[j0 + a] = kSel; return 0; }
      return 1;
    };
#pragma unroll
    for (int a = 0; a < PER; ++a) visit(a);
#pragma unroll
    for (int a = PER - 1; a >= 0; --a) undecided |= visit(a);
    if (!__syncthreads_or(undecided)) break;
  }
}
template <bool PLANE>
__device__ __forceinline__ void nms_fixed_point_any(volatile uint32_t* w, int np, int hw)
{
  // (block-uniform choice: the smallest chunk that covers the ring with the block's threads)
  if (np <= (int)blockDim.x) nms_fixed_point<PLANE, 1>(w, np, hw);
  else if (np <= 2 * (int)blockDim.x) nms_fixed_point<PLANE, 2>(w, np, hw);
  else if (np <= 4 * (int)blockDim.x) nms_fixed_point<PLANE, 4>(w, np, hw);
  else nms_fixed_point<PLANE, 8>(w, np, hw);
}

// after a selection round: label the winners, clear the validity bit `vbit` within +-hw of them
__device__ void nms_apply(const uint32_t* w, uint8_t* flags, int np, int hw, uint8_t vbit, uint8_t lbit)
{
  for (int j = threadIdx.x; j < np; j += blockDim.x)
  {
    const int b = max(0, j - hw), e = min(np - 1, j + hw);
    bool near = false;
    for (int k = b; k <= e; ++k) near |= (w[k] == kSel);
    uint8_t f = flags[j];
    if (w[j] == kSel) f |= lbit;
    if (near) f &= ~vbit;
    flags[j] = f;
  }
  __syncthreads();
}

__global__ __launch_bounds__(kLabelThreads) void k_label(const float* __restrict__ g_angle, const float* __restrict__ g_gap,
                                               const float* __restrict__ g_sal, const float* __restrict__ g_int,
                                               const int* __restrict__ ring_start, const int* __restrict__ ring_len,
                                               int* __restrict__ ring_meta, ExtractConst c, uint8_t* __restrict__ valid,
                                               uint8_t* __restrict__ label, int* __restrict__ ring_counts)
{
  static_assert(8 * kLabelThreads >= kMaxRingPoints, "a thread owns at most 8 points of its ring");
  __shared__ uint32_t wd[kMaxRingPoints];    // state + priority of the selection under way (nms_fixed_point)
  __shared__ uint8_t flags[kMaxRingPoints];  // bits 0-2 validity E/P/B, bits 3-5 label E/P/B
  __shared__ int cnt[3];
  const int r = blockIdx.x;
  if (r >= ring_meta[0]) return;
  const int s0 = ring_start[r];
  const int np = ring_len[r];
  if (threadIdx.x < 3) cnt[threadIdx.x] = 0;
  if (np > kMaxRingPoints)
  {
    if (threadIdx.x == 0) atomicOr(&ring_meta[2], 2);
    return;
  }
  const int W = c.W;
  if (np < 2 * W + 1)
  {
    // IsScanLineAlmostEmpty: nothing is labelled, validity stays cleared (SSKE.cxx:487-490)
    for (int j = threadIdx.x; j < np; j += blockDim.x) label[s0 + j] = 0;
    if (threadIdx.x < 3) ring_counts[r * 3 + threadIdx.x] = 0;
    return;
  }
  for (int j = threadIdx.x; j < np; j += blockDim.x) flags[j] = valid[s0 + j] & 7;
  __syncthreads();

  // --- edges: depth gap, angle, saliency, intensity gap, in this order (SSKE.cxx:526-533)
  for (int crit = 0; crit < 4; ++crit)
  {
    const float* src = crit == 0 ? g_gap : crit == 1 ? g_angle : crit == 2 ? g_sal : g_int;
    const float thr = crit == 0 ? c.sq_depth_gap_thr : crit == 1 ? c.edge_angle_thr : crit == 2 ? c.sq_saliency_thr : c.intensity_thr;
    const int hw = crit == 1 ? W : crit == 3 ? 1 : W - 1;
    for (int j = threadIdx.x; j < np; j += blockDim.x)
    {
      const float v = src[s0 + j];
      wd[j] = ((v >= thr) && (flags[j] & 1)) ? nms_word<false>(v) : 0u;  // NaN never qualifies
    }
    __syncthreads();
    nms_fixed_point_any<false>(wd, np, hw);
    nms_apply(wd, flags, np, hw, 1, 8);
  }

  // --- planes: ascending sin angle, skip < 1e-6, stop above the threshold, +-4 window (SSKE.cxx:536-563)
  {
    const int hw = 4;
    for (int j = threadIdx.x; j < np; j += blockDim.x)
    {
      const float v = g_angle[s0 + j];
      const bool cand = (flags[j] & 2) && !((double)v < 1e-6) && (v <= c.plane_thr);  // NaN fails v <= thr
      wd[j] = cand ? nms_word<true>(v) : 0u;
    }
    __syncthreads();
    nms_fixed_point_any<true>(wd, np, hw);
    nms_apply(wd, flags, np, hw, 2, 16);
  }

  // --- blobs (SSKE.cxx:568-572) + validity bit set back for labelled points (:584) + counts
  int ce = 0, cp = 0, cb = 0;
  for (int j = threadIdx.x; j < np; j += blockDim.x)
  {
    uint8_t f = flags[j];
    if ((j % 3) == 0 && (f & 4)) f |= 32;
    const uint8_t lab = (f >> 3) & 7;
    label[s0 + j] = lab;
    valid[s0 + j] = (f & 7) | lab;
    ce += lab & 1; cp += (lab >> 1) & 1; cb += (lab >> 2) & 1;
  }
  for (int o = 32; o > 0; o >>= 1)
  {
    ce += __shfl_down(ce, o); cp += __shfl_down(cp, o); cb += __shfl_down(cb, o);
  }
  if ((threadIdx.x & 63) == 0)
  {
    atomicAdd(&cnt[0], ce); atomicAdd(&cnt[1], cp); atomicAdd(&cnt[2], cb);
  }
  __syncthreads();
  if (threadIdx.x < 3) ring_counts[r * 3 + threadIdx.x] = cnt[threadIdx.x];
}

// --------------------------------------------------------------------------------------------
constexpr int kCompactThreads = 1024;  // two points of a 2 048-point ring per thread: the dependent loads (label -> original index -> point) side by side
__global__ __launch_bounds__(kCompactThreads) void k_compact(const float4* __restrict__ frame, const uint32_t* __restrict__ orig,
                                                 const uint8_t* __restrict__ label, const int* __restrict__ ring_start,
                                                 const int* __restrict__ ring_len, const int* __restrict__ ring_meta,
                                                 const int* __restrict__ ring_counts, float4* __restrict__ out_e,
                                                 float4* __restrict__ out_p, float4* __restrict__ out_b, int* __restrict__ kp_count,
                                                 unsigned type_mask, unsigned long long* __restrict__ time_range)
{
  constexpr int kWaves = kCompactThreads / 64;
  __shared__ int base[3];
  __shared__ unsigned long long wtot[kWaves];
  __shared__ unsigned long long tlo[kWaves], thi[kWaves];
  const int r = blockIdx.x;
  const int nr = ring_meta[0];
  if (r >= nr) return;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (wave < 3)
  {
    // keypoints of type `wave` in the rings in front of this one: the lanes share the rings, one reduction
    int s = 0;
    for (int q = lane; q < r; q += 64) s += ring_counts[q * 3 + wave];
    for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o);
    if (lane == 0)
    {
      base[wave] = s;
      // a type the caller does not use (Slam::UseKeypoints) is dropped here: count 0, nothing written
      if (r == nr - 1) kp_count[wave] = ((type_mask >> wave) & 1u) ? s + ring_counts[r * 3 + wave] : 0;
    }
  }
  const int s0 = ring_start[r];
  const int np = ring_len[r];
  const int per = (np + blockDim.x - 1) / blockDim.x;
  const int jb = min(np, (int)threadIdx.x * per), je = min(np, jb + per);
  unsigned long long mine = 0;
  for (int j = jb; j < je; ++j)
  {
    const uint8_t l = label[s0 + j] & type_mask;
    mine += (unsigned long long)(l & 1) | ((unsigned long long)((l >> 1) & 1) << 21) | ((unsigned long long)((l >> 2) & 1) << 42);
  }
  // exclusive scan of the (packed) counts over the block: inside the wavefront by shuffles, across by the wavefronts' totals
  unsigned long long inc = mine;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1)
  {
    const unsigned long long t = __shfl_up(inc, o);
    if (lane >= o) inc += t;
  }
  if (lane == 63) wtot[wave] = inc;
  __syncthreads();
  unsigned long long before = 0;
  for (int w = 0; w < wave; ++w) before += wtot[w];
  unsigned long long excl = before + inc - mine;
  int pe = base[0] + (int)(excl & 0x1fffff);
  int pp = base[1] + (int)((excl >> 21) & 0x1fffff);
  int pb = base[2] + (int)((excl >> 42) & 0x1fffff);
  // the keypoints' time range (Slam::InitUndistortion, Slam.cxx:1291-1300) is reduced on the way
  unsigned long long lo = ~0ull, hi = 0ull;
  for (int j = jb; j < je; ++j)
  {
    const uint8_t l = label[s0 + j] & type_mask;
    if (!l) continue;
    const size_t o = orig[s0 + j];
    const float4 a = frame[2 * o], b = frame[2 * o + 1];
    if (l & 1) { out_e[2 * (size_t)pe] = a; out_e[2 * (size_t)pe + 1] = b; ++pe; }
    if (l & 2) { out_p[2 * (size_t)pp] = a; out_p[2 * (size_t)pp + 1] = b; ++pp; }
    if (l & 4) { out_b[2 * (size_t)pb] = a; out_b[2 * (size_t)pb + 1] = b; ++pb; }
    const unsigned long long tb = time_bits(b);
    lo = tb < lo ? tb : lo;
    hi = tb > hi ? tb : hi;
  }
  for (int s = 32; s > 0; s >>= 1)
  {
    const unsigned long long l2 = __shfl_down(lo, s), h2 = __shfl_down(hi, s);
    lo = l2 < lo ? l2 : lo;
    hi = h2 > hi ? h2 : hi;
  }
  if ((threadIdx.x & 63) == 0) { tlo[threadIdx.x >> 6] = lo; thi[threadIdx.x >> 6] = hi; }
  __syncthreads();
  if (threadIdx.x == 0)
  {
    for (int w = 1; w < kWaves; ++w) { lo = tlo[w] < lo ? tlo[w] : lo; hi = thi[w] > hi ? thi[w] : hi; }
    if (hi >= lo) { atomicMin(&time_range[0], lo); atomicMax(&time_range[1], hi); }
  }
}

__global__ void k_debug_gather(const uint32_t* __restrict__ orig, int n, int id, const float* __restrict__ src,
                               const uint8_t* __restrict__ bits, int bit, float* __restrict__ out)
{
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  out[orig[j]] = (id < 4) ? src[j] : (float)((bits[j] >> bit) & 1);
}

__global__ void k_transform_points(float4* __restrict__ pts, int n, Rigid T, double time_offset, int rigid)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float4 a = pts[2 * (size_t)i];
  float4 b = pts[2 * (size_t)i + 1];
  if (rigid)  // AggregateFrames leaves the coordinates alone when the transform is the identity (Slam.cxx:1557-1562)
  {
    double ox, oy, oz;
    rigid_apply(T, (double)a.x, (double)a.y, (double)a.z, ox, oy, oz);
    a.x = (float)ox; a.y = (float)oy; a.z = (float)oz;
  }
  double t = __hiloint2double(__float_as_int(b.y), __float_as_int(b.x)) + time_offset;
  b.x = __int_as_float(__double2loint(t));
  b.y = __int_as_float(__double2hiint(t));
  pts[2 * (size_t)i] = a;
  pts[2 * (size_t)i + 1] = b;
}

ExtractConst make_const(const lsa_extract_params_t* p, float az_res)
{
  // host-side constants of InvalidateNotUsablePoints / LineFitting, evaluated with the same libm
  // float functions the reference uses (SSKE.cxx:52-55, 90, 210-222, 313, 476-477)
  ExtractConst c;
  c.W = p->neighbor_width;
  c.min_dist = p->min_distance_to_sensor;
  const float angleBeamNormal = float((90 - p->min_beam_surface_angle) / 180. * M_PI);
  float az = az_res;
  if (az < 1e-6 || M_PI / 4 < az) az = float(0.2 / 180. * M_PI);
  c.max_pos_diff_coeff = std::sin(az) / std::cos(az + angleBeamNormal);
  const float lineMaxAngle = float(40. * 0.017453293);
  c.line_max_sin = std::sin(lineMaxAngle);
  const float lineMaxDist = 0.02f;
  c.line_sq_max_dist = lineMaxDist * lineMaxDist;
  c.sq_dist_to_line_thr = p->dist_to_line_threshold * p->dist_to_line_threshold;
  c.plane_thr = p->plane_sin_angle_threshold;
  c.edge_angle_thr = p->edge_sin_angle_threshold;
  c.sq_depth_gap_thr = p->edge_depth_gap_threshold * p->edge_depth_gap_threshold;
  c.sq_saliency_thr = p->edge_saliency_threshold * p->edge_saliency_threshold;
  c.intensity_thr = p->edge_intensity_gap_threshold;
  return c;
}

}  // namespace

extern "C" {

// The extraction kernels of one frame on `st`: ring bucketing, validity, curvature scores, labels, compaction into
// kp_out[type]; `out` receives the 64-byte summary (counts, ring meta, time range of the keypoints).  The per-point
// scratch buffers of the context are shared by every caller: one frame at a time.
static void enqueue_extract(lsa_ctx* ctx, const float4* frame4, int n, const ExtractConst& c, hipStream_t st, int* out, lsa_point_t* const kp_out[3],
                            unsigned type_mask)
{
  int* ring_meta = out + 4;
  const int nblocks = (n + kBucketChunk - 1) / kBucketChunk;
  hipLaunchKernelGGL(k_extract_init, dim3(1), dim3(64), 0, st, out);
  {
    ProfScope ps(ctx, "ring_bucket", (double)n * (4 + 32 + 16 + 4 + 2 + 1), st);
    hipLaunchKernelGGL(k_ring_hist, dim3(nblocks), dim3(256), 0, st, frame4, n, ctx->block_hist, ring_meta);
    hipLaunchKernelGGL(k_ring_scan, dim3(kMaxRings), dim3(64), 0, st, ctx->block_hist, nblocks, ctx->ring_len, ring_meta);
    hipLaunchKernelGGL(k_ring_scatter, dim3(nblocks), dim3(256), 0, st, frame4, n, ctx->block_hist, ctx->ring_len, ctx->ring_start,
                       ctx->xyzi, ctx->orig, ctx->ring_of, ctx->valid);
  }
  {
    ProfScope ps(ctx, "invalidate", (double)n * (16 + 2 + 1), st);
    hipLaunchKernelGGL(k_invalidate, dim3((n + 255) / 256), dim3(256), 0, st, ctx->xyzi, ctx->ring_of, ctx->ring_start, ctx->ring_len, n, c,
                       ctx->valid);
  }
  {
    ProfScope ps(ctx, "curvature", (double)n * (16 + 2 + 1 + 16), st);
    if (c.W <= 4)
      hipLaunchKernelGGL(k_curvature<4>, dim3((n + 127) / 128), dim3(128), 0, st, ctx->xyzi, ctx->ring_of, ctx->ring_start, ctx->ring_len, n, c,
                         ctx->valid, ctx->score[0], ctx->score[1], ctx->score[2], ctx->score[3]);
    else
      hipLaunchKernelGGL(k_curvature<8>, dim3((n + 127) / 128), dim3(128), 0, st, ctx->xyzi, ctx->ring_of, ctx->ring_start, ctx->ring_len, n, c,
                         ctx->valid, ctx->score[0], ctx->score[1], ctx->score[2], ctx->score[3]);
  }
  {
    ProfScope ps(ctx, "label_nms", (double)n * (16 + 1 + 1 + 1), st);
    hipLaunchKernelGGL(k_label, dim3(kMaxRings), dim3(kLabelThreads), 0, st, ctx->score[0], ctx->score[1], ctx->score[2], ctx->score[3], ctx->ring_start,
                       ctx->ring_len, ring_meta, c, ctx->valid, ctx->label, ctx->ring_counts);
  }
  {
    ProfScope ps(ctx, "compact", (double)n * (1 + 4), st);
    hipLaunchKernelGGL(k_compact, dim3(kMaxRings), dim3(kCompactThreads), 0, st, frame4, ctx->orig, ctx->label, ctx->ring_start, ctx->ring_len,
                       ring_meta, ctx->ring_counts, reinterpret_cast<float4*>(kp_out[0]),
                       reinterpret_cast<float4*>(kp_out[1]), reinterpret_cast<float4*>(kp_out[2]),
                       out, type_mask, reinterpret_cast<unsigned long long*>(out + 12));
  }
}

// One device frame through the extraction kernels.  append == false: Slam::ExtractKeypoints' first frame (the
// current keypoints become the previous ones, Slam.cxx:751); append == true: a further frame of the same
// AddFrames call, whose keypoints go behind those already there (AggregateFrames, Slam.cxx:1512-1578).
static int extract_frame(lsa_ctx* ctx, const lsa_extract_params_t* params, int counts[3], bool append, const double* base_to_lidar, double time_offset)
{
  if (!ctx || !params || !counts) return ctx ? ctx->fail(LSA_E_ARG, "lsa_extract_keypoints: null argument") : LSA_E_ARG;
  if (!ctx->frame || ctx->frame_n <= 0) return ctx->fail(LSA_E_STATE, "lsa_extract_keypoints: no frame uploaded");
  if (params->neighbor_width < 1 || params->neighbor_width > 8)
    return ctx->fail(LSA_E_ARG, "lsa_extract_keypoints: NeighborWidth must be in [1, 8]");
  LSA_HIP(ctx, hipSetDevice(ctx->device));
  const int n = ctx->frame_n;
  int base[3] = {0, 0, 0};
  if (append)
    for (int k = 0; k < 3; ++k) base[k] = ctx->kp_n[LSA_SET_RAW_CURRENT][k];
  // a type's keypoints of this frame are at most its points: room for them behind what is there
  int rc = ensure_capacity(ctx, n + std::max(base[0], std::max(base[1], base[2])));
  if (rc) return rc;
  hipStream_t st = ctx->stream;
  const ExtractConst c = make_const(params, ctx->az_res);

  // a look-ahead extraction (lsa_extract_prefetch) of exactly this frame with exactly these settings is adopted;
  // any other one has to finish first, because the per-point scratch buffers are shared
  bool adopt = false;
  if (ctx->prefetch_pending)
  {
    adopt = !append && !base_to_lidar && time_offset == 0. && ctx->prefetch_frame == ctx->frame && ctx->prefetch_n == n &&
            std::memcmp(&ctx->prefetch_params, params, sizeof(*params)) == 0 && ctx->prefetch_az == ctx->az_res &&
            ctx->prefetch_mask == ctx->kp_type_mask;
    if (adopt) LSA_HIP(ctx, hipEventSynchronize(ctx->ev_prefetch));
    else LSA_HIP(ctx, hipStreamSynchronize(ctx->prefetch_stream));
    ctx->prefetch_pending = false;
  }
  if (!append)
  {
    // Slam::ExtractKeypoints: current keypoints become the previous ones (Slam.cxx:751)
    for (int k = 0; k < 3; ++k)
    {
      std::swap(ctx->kp[LSA_SET_RAW_CURRENT][k], ctx->kp[LSA_SET_RAW_PREVIOUS][k]);
      ctx->kp_ver[LSA_SET_RAW_PREVIOUS][k] = ctx->kp_ver[LSA_SET_RAW_CURRENT][k];
      ctx->kp_n[LSA_SET_RAW_PREVIOUS][k] = ctx->kp_n[LSA_SET_RAW_CURRENT][k];
      ctx->kp_n[LSA_SET_RAW_CURRENT][k] = 0;
    }
    ctx->kp_time_valid[LSA_SET_RAW_PREVIOUS] = ctx->kp_time_valid[LSA_SET_RAW_CURRENT];
    ctx->kp_time[LSA_SET_RAW_PREVIOUS][0] = ctx->kp_time[LSA_SET_RAW_CURRENT][0];
    ctx->kp_time[LSA_SET_RAW_PREVIOUS][1] = ctx->kp_time[LSA_SET_RAW_CURRENT][1];
  }
  ctx->kp_time_valid[LSA_SET_RAW_CURRENT] = false;

  int* hp = reinterpret_cast<int*>(ctx->host_pinned);
  if (adopt)
  {
    // the keypoints are there already: the look-ahead buffers become the current sets, their summary was copied
    // to the host when the kernels finished
    for (int k = 0; k < 3; ++k) std::swap(ctx->kp[LSA_SET_RAW_CURRENT][k], ctx->kp_next[k]);
    hp = ctx->host_next;
    ctx->prefetch_adopted++;
  }
  else
  {
    lsa_point_t* kp_out[3];
    for (int k = 0; k < 3; ++k) kp_out[k] = ctx->kp[LSA_SET_RAW_CURRENT][k] + base[k];
    enqueue_extract(ctx, reinterpret_cast<const float4*>(ctx->frame), n, c, st, ctx->extract_out, kp_out, ctx->kp_type_mask);
    // counts, ring_meta and the keypoints' time range (Slam::InitUndistortion needs it later) in one 64-byte read-back
    LSA_HIP(ctx, hipMemcpyAsync(hp, ctx->extract_out, 16 * sizeof(int), hipMemcpyDeviceToHost, st));
    LSA_HIP(ctx, hipStreamSynchronize(st));
  }
  unsigned long long hpt[2];
  std::memcpy(hpt, hp + 12, sizeof(hpt));
  if (hp[6] & 1) return ctx->fail(LSA_E_CAPACITY, "lsa_extract_keypoints: laser_id >= 512 is not supported");
  if (hp[6] & 2) return ctx->fail(LSA_E_CAPACITY, "lsa_extract_keypoints: more than 8192 points on one laser ring");
  ctx->nb_rings_seen = std::max(ctx->nb_rings_seen, hp[4]);
  for (int k = 0; k < 3; ++k)
  {
    counts[k] = hp[k];
    ctx->kp_n[LSA_SET_RAW_CURRENT][k] = base[k] + hp[k];
    ctx->kp_ver[LSA_SET_RAW_CURRENT][k] = ++ctx->kp_clock;
  }
  if (!append && !base_to_lidar && time_offset == 0.)
  {
    finish_time_range(ctx, LSA_SET_RAW_CURRENT, hpt);
    return LSA_OK;
  }
  // AggregateFrames(keypoints, false) on the points just written: time offset to the first frame's stamp and the
  // sensor's pose in BASE (Slam.cxx:1536-1575); the set's time range is reduced again when it is asked for
  Rigid R;
  const double identity[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
  row_major_to_rt(base_to_lidar ? base_to_lidar : identity, R.R, R.t);
  for (int k = 0; k < 3; ++k)
    if (hp[k] > 0 && (base_to_lidar || time_offset != 0.))
      hipLaunchKernelGGL(k_transform_points, dim3((hp[k] + 255) / 256), dim3(256), 0, st,
                         reinterpret_cast<float4*>(ctx->kp[LSA_SET_RAW_CURRENT][k] + base[k]), hp[k], R, time_offset, base_to_lidar ? 1 : 0);
  return LSA_OK;
}

// ---- SpinningFrameAdvancementEstimator on the device (lidar_conversions/src/Utilities.h:62-114) --------------------
// The estimator walks the frame in arrival order and keeps, per laser ring, the advancement of the ring's previous point:
// a point whose advancement (in [0, 1)) is lower than its ring's previous one has passed the turn and gets + 1 -- and
// from then on so does every later point of that ring (their advancement < 1 <= the stored value).  Per ring that is
// "the first descent and everything after it": with the frame bucketed by ring (arrival order kept inside a ring) the
// first descent is one comparison with the ring-major predecessor and a minimum per ring.
__global__ __launch_bounds__(256) void k_first_wrap(const float4* __restrict__ frame, const uint32_t* __restrict__ orig, const uint16_t* __restrict__ ring_of,
                                                    const int* __restrict__ ring_start, int n, int* __restrict__ first_wrap)
{
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  const int ring = ring_of[p];
  if (p == ring_start[ring]) return;  // the ring's first point compares with 0.0: never lower
  const float4 b = frame[2 * (size_t)orig[p] + 1], a = frame[2 * (size_t)orig[p - 1] + 1];
  const double cur = __hiloint2double(__float_as_int(b.y), __float_as_int(b.x)), prev = __hiloint2double(__float_as_int(a.y), __float_as_int(a.x));
  if (cur < prev) atomicMin(&first_wrap[ring], p - ring_start[ring]);
}
__global__ __launch_bounds__(256) void k_time_from_advancement(float4* __restrict__ frame, const uint32_t* __restrict__ orig, const uint16_t* __restrict__ ring_of,
                                                               const int* __restrict__ ring_start, int n, const int* __restrict__ first_wrap, double rpm,
                                                               int first_packet)
{
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  const int ring = ring_of[p];
  float4 b = frame[2 * (size_t)orig[p] + 1];
  double adv = __hiloint2double(__float_as_int(b.y), __float_as_int(b.x));
  if (p - ring_start[ring] >= first_wrap[ring]) adv += 1.;
  const double t = (first_packet ? adv : adv - 1) / rpm * 60.;  // VelodyneToLidarNode.cxx:106
  const long long tb = __double_as_longlong(t);
  b.x = __int_as_float((int)(tb & 0xffffffffll));
  b.y = __int_as_float((int)(tb >> 32));
  frame[2 * (size_t)orig[p] + 1] = b;
}

}  // extern "C"

namespace lsa
{
// frame (AoS, n points on the device) holds in its time field the advancement of every point in [0, 1): replaces it by
// the estimator's time.  Returns LSA_E_CAPACITY when a laser id cannot be bucketed (>= 512 rings): the caller converts
// on the host then.
int time_from_advancement(lsa_ctx* ctx, lsa_point_t* frame, int n, double rpm, int first_packet)
{
  hipStream_t st = ctx->stream;
  if (ctx->prefetch_pending)
  {
    LSA_HIP(ctx, hipStreamSynchronize(ctx->prefetch_stream));  // the bucketing buffers are shared with the look-ahead extraction
  }
  float4* f4 = reinterpret_cast<float4*>(frame);
  const int nblocks = (n + kBucketChunk - 1) / kBucketChunk;
  int* meta = ctx->extract_out + 4;
  hipLaunchKernelGGL(k_extract_init, dim3(1), dim3(64), 0, st, ctx->extract_out);
  hipLaunchKernelGGL(k_ring_hist, dim3(nblocks), dim3(256), 0, st, f4, n, ctx->block_hist, meta);
  hipLaunchKernelGGL(k_ring_scan, dim3(kMaxRings), dim3(64), 0, st, ctx->block_hist, nblocks, ctx->ring_len, meta);
  hipLaunchKernelGGL(k_ring_scatter, dim3(nblocks), dim3(256), 0, st, f4, n, ctx->block_hist, ctx->ring_len, ctx->ring_start, ctx->xyzi, ctx->orig,
                     ctx->ring_of, ctx->valid);
  int* first_wrap = ctx->ring_counts;  // [kMaxRings] of its 3 x kMaxRings ints
  LSA_HIP(ctx, hipMemsetAsync(first_wrap, 0x7f, kMaxRings * sizeof(int), st));
  hipLaunchKernelGGL(k_first_wrap, dim3((n + 255) / 256), dim3(256), 0, st, f4, ctx->orig, ctx->ring_of, ctx->ring_start, n, first_wrap);
  hipLaunchKernelGGL(k_time_from_advancement, dim3((n + 255) / 256), dim3(256), 0, st, f4, ctx->orig, ctx->ring_of, ctx->ring_start, n, first_wrap, rpm,
                     first_packet);
  int* hp = reinterpret_cast<int*>(ctx->host_pinned);
  LSA_HIP(ctx, hipMemcpyAsync(hp, ctx->extract_out, 16 * sizeof(int), hipMemcpyDeviceToHost, st));
  LSA_HIP(ctx, hipStreamSynchronize(st));
  if (hp[6] & 3) return LSA_E_CAPACITY;
  return LSA_OK;
}
}  // namespace lsa

extern "C" {

int lsa_extract_keypoints(lsa_ctx* ctx, const lsa_extract_params_t* params, int counts[3])
{
  return extract_frame(ctx, params, counts, false, nullptr, 0.);
}

// look-ahead extraction of `frame` (n points) on the look-ahead stream, behind `after` when given
static int prefetch_enqueue(lsa_ctx* ctx, const lsa_point_t* frame, int n, const lsa_extract_params_t* params, hipEvent_t after)
{
  if (params->neighbor_width < 1 || params->neighbor_width > 8) return ctx->fail(LSA_E_ARG, "lsa_extract_prefetch: NeighborWidth must be in [1, 8]");
  LSA_HIP(ctx, hipSetDevice(ctx->device));
  if (ctx->prefetch_pending)
  {
    LSA_HIP(ctx, hipStreamSynchronize(ctx->prefetch_stream));  // one look-ahead at a time
    ctx->prefetch_pending = false;
  }
  int rc = ensure_capacity(ctx, n);
  if (rc) return rc;
  if (!ctx->kp_next[0])
    for (int k = 0; k < 3; ++k) LSA_HIP(ctx, hipMalloc((void**)&ctx->kp_next[k], (size_t)ctx->cap_n * sizeof(lsa_point_t)));
  if (after) LSA_HIP(ctx, hipStreamWaitEvent(ctx->prefetch_stream, after, 0));
  // the scratch buffers are free: the extraction of the current frame ended with a read-back on ctx->stream
  const ExtractConst c = make_const(params, ctx->az_res);
  enqueue_extract(ctx, reinterpret_cast<const float4*>(frame), n, c, ctx->prefetch_stream, ctx->extract_out_next, ctx->kp_next, ctx->kp_type_mask);
  LSA_HIP(ctx, hipMemcpyAsync(ctx->host_next, ctx->extract_out_next, 16 * sizeof(int), hipMemcpyDeviceToHost, ctx->prefetch_stream));
  LSA_HIP(ctx, hipEventRecord(ctx->ev_prefetch, ctx->prefetch_stream));
  ctx->prefetch_pending = true;
  ctx->prefetch_frame = frame;
  ctx->prefetch_n = n;
  ctx->prefetch_params = *params;
  ctx->prefetch_az = ctx->az_res;
  ctx->prefetch_mask = ctx->kp_type_mask;
  return LSA_OK;
}

int lsa_extract_prefetch(lsa_ctx* ctx, int slot, const lsa_extract_params_t* params)
{
  if (!ctx || !params) return ctx ? ctx->fail(LSA_E_ARG, "lsa_extract_prefetch: null argument") : LSA_E_ARG;
  if (slot < 0 || slot >= (int)ctx->store.size() || !ctx->store[slot].first) return ctx->fail(LSA_E_ARG, "lsa_extract_prefetch: empty slot");
  return prefetch_enqueue(ctx, ctx->store[slot].first, ctx->store[slot].second, params, nullptr);
}

int lsa_extract_prefetch_uploaded(lsa_ctx* ctx, const lsa_extract_params_t* params)
{
  if (!ctx || !params) return ctx ? ctx->fail(LSA_E_ARG, "lsa_extract_prefetch_uploaded: null argument") : LSA_E_ARG;
  if (ctx->inbox_queue.empty()) return ctx->fail(LSA_E_STATE, "lsa_extract_prefetch_uploaded: no frame uploaded ahead");
  lsa::FrameInbox& in = ctx->inbox[ctx->inbox_queue.front()];
  if (in.state.load(std::memory_order_acquire) != 2) return ctx->fail(LSA_E_STATE, "lsa_extract_prefetch_uploaded: the upload has not been enqueued yet");
  // the azimuthal resolution has to be known: it is estimated from the first frame when that frame is handed over
  if (ctx->az_res < 1e-6f) return ctx->fail(LSA_E_STATE, "lsa_extract_prefetch_uploaded: azimuthal resolution not estimated yet");
  return prefetch_enqueue(ctx, in.dev, in.n, params, in.ev);
}

int lsa_extract_prefetch_adopted(const lsa_ctx* ctx) { return ctx ? ctx->prefetch_adopted : 0; }

int lsa_extract_keypoints_more(lsa_ctx* ctx, const lsa_extract_params_t* params, const double base_to_lidar[16], double time_offset, int counts[3])
{
  return extract_frame(ctx, params, counts, true, base_to_lidar, time_offset);
}

int lsa_keypoint_count(const lsa_ctx* ctx, int set, int type)
{
  if (!ctx || set < 0 || set > 2 || type < 0 || type > 2) return LSA_E_ARG;
  return ctx->kp_n[set][type];
}

int lsa_download_keypoints(lsa_ctx* ctx, int set, int type, lsa_point_t* out, int capacity)
{
  if (!ctx || set < 0 || set > 2 || type < 0 || type > 2 || (!out && capacity > 0)) return ctx ? ctx->fail(LSA_E_ARG, "lsa_download_keypoints: bad argument") : LSA_E_ARG;
  const int n = std::min(capacity, ctx->kp_n[set][type]);
  if (n <= 0) return 0;
  LSA_HIP(ctx, hipMemcpyAsync(out, ctx->kp[set][type], (size_t)n * sizeof(lsa_point_t), hipMemcpyDeviceToHost, ctx->stream));
  LSA_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return n;
}

int lsa_set_keypoint_types(lsa_ctx* ctx, unsigned type_mask)
{
  if (!ctx || (type_mask & ~7u)) return LSA_E_ARG;
  ctx->kp_type_mask = type_mask;
  return LSA_OK;
}

int lsa_set_keypoints(lsa_ctx* ctx, int set, int type, const lsa_point_t* pts, int k)
{
  if (!ctx || set < 0 || set > 2 || type < 0 || type > 2 || k < 0 || (!pts && k > 0)) return ctx ? ctx->fail(LSA_E_ARG, "lsa_set_keypoints: bad argument") : LSA_E_ARG;
  LSA_HIP(ctx, hipSetDevice(ctx->device));
  int rc = ensure_capacity(ctx, k);
  if (rc) return rc;
  if (k > 0)
  {
    LSA_HIP(ctx, hipMemcpyAsync(ctx->kp[set][type], pts, (size_t)k * sizeof(lsa_point_t), hipMemcpyHostToDevice, ctx->stream));
    LSA_HIP(ctx, hipStreamSynchronize(ctx->stream));  // pts may be pageable and reused by the caller
  }
  ctx->kp_n[set][type] = k;
  ctx->kp_ver[set][type] = ++ctx->kp_clock;
  ctx->kp_time_valid[set] = false;
  return LSA_OK;
}

int lsa_download_debug(lsa_ctx* ctx, int array_id, float* out, int capacity)
{
  if (!ctx || !out || array_id < 0 || array_id > 9) return ctx ? ctx->fail(LSA_E_ARG, "lsa_download_debug: bad argument") : LSA_E_ARG;
  if (ctx->prefetch_pending) return ctx->fail(LSA_E_STATE, "lsa_download_debug: the per-point arrays are in use by a look-ahead extraction (lsa_extract_prefetch)");
  const int n = ctx->frame_n;
  if (capacity < n) return ctx->fail(LSA_E_CAPACITY, "lsa_download_debug: capacity < frame size");
  int rc = ensure_scratch(ctx, (size_t)n * sizeof(float));
  if (rc) return rc;
  // reference names: 0 sin_angle(Angles) 1 saliency 2 depth_gap 3 intensity_gap
  const float* src = array_id == 0 ? ctx->score[0] : array_id == 1 ? ctx->score[2] : array_id == 2 ? ctx->score[1] : ctx->score[3];
  const uint8_t* bits = array_id < 7 ? ctx->label : ctx->valid;
  const int bit = array_id < 4 ? 0 : array_id < 7 ? array_id - 4 : array_id - 7;
  hipLaunchKernelGGL(k_debug_gather, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, ctx->orig, n, array_id, src, bits, bit,
                     (float*)ctx->scratch_out);
  LSA_HIP(ctx, hipMemcpyAsync(out, ctx->scratch_out, (size_t)n * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
  LSA_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return n;
}

int lsa_transform_keypoints(lsa_ctx* ctx, int set, int type, const double T[16], double time_offset)
{
  if (!ctx || !T || set < 0 || set > 2 || type < 0 || type > 2) return ctx ? ctx->fail(LSA_E_ARG, "lsa_transform_keypoints: bad argument") : LSA_E_ARG;
  const int n = ctx->kp_n[set][type];
  if (time_offset != 0.) ctx->kp_time_valid[set] = false;
  ctx->kp_ver[set][type] = ++ctx->kp_clock;
  if (n <= 0) return LSA_OK;
  Rigid R;
  row_major_to_rt(T, R.R, R.t);
  ProfScope ps(ctx, "transform_keypoints", (double)n * 64);
  hipLaunchKernelGGL(k_transform_points, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, reinterpret_cast<float4*>(ctx->kp[set][type]), n, R,
                     time_offset, 1);
  return LSA_OK;
}

}  // extern "C"
