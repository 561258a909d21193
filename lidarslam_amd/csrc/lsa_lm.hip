// lsa_lm.hip -- LocalOptimizer::Solve (slam_lib/src/LocalOptimizer.cxx:74-102) as ONE launch.
//
// The host-driven loop (host/lsa_lm.cpp) pays a kernel launch and a PCIe round trip per trust-region
// evaluation, ~4 per solve, ~25 per frame.  Here the whole Levenberg-Marquardt loop of Ceres (unpinned
// master >= 2.0 in the reference CI: TukeyLoss a^2/3, ScaledLoss(weight), DENSE_QR on a 6-column
// Jacobian = the 6x6 normal equations, Solver::Options defaults) runs inside one kernel:
//
//   * every block evaluates its share of the residual blocks at the candidate pose (lsa_accum.h) and reduces
//     it in a fixed order to 29 sums;
//   * the blocks exchange those sums through 8-byte {tag, half of a double} granules in device memory, each
//     ONE relaxed agent-scope atomic store / load (MI355X: `sc1`, write-through past the non-coherent per-XCD
//     L2s): tag and payload are one memory object, so no fence, flag or ordering between locations is needed.
//     Two parities of slots: a block can be at most one evaluation ahead of the slowest;
//   * EVERY block folds all partial sums in block order and runs the 6x6 trust-region algebra itself -- same
//     inputs, same instructions, same result in every block -- so the next candidate needs no broadcast;
//   * block 0 hands pose, summary and the normal equations at the final point to the host through the same
//     kind of granules in coherent host memory.
// The arithmetic of the trust-region step is the host loop's, operation for operation (no FMA contraction, IEEE
// division and square root): given the same sums both take the same decisions.  The sums differ in the last
// bits (more blocks, and sin/cos of the angles come from lsa_pmath.h instead of libm).
// Every spin is bounded: a block that waits too long gives up, the launch drains, the host falls back to the
// host-driven loop and counts it (lsa_ctx::lm_fallbacks).
#include <chrono>
#include <cmath>
#include "lsa_accum.h"
#include "lsa_device_math.h"
#include "../../include/lsa_pmath.h"

using namespace lsa;

namespace
{

typedef unsigned long long u64;
#ifndef LSA_RECORD_PREFETCH
#define LSA_RECORD_PREFETCH 1
#endif
constexpr bool kRecordPrefetch = LSA_RECORD_PREFETCH != 0;  // (0: A/B builds)

struct LmParams
{
  RecordSet set;
  double x0[6];
  int two_d;
  int max_iter;
  int min_matches;
  unsigned tag_base;
  int cslots;  // residual blocks per thread kept in LDS between the evaluations
  int give_up_block;    // test hook: this workgroup abandons the exchange at its first evaluation (-1: none)
  const IcpGate* gate;  // not null: the launch was enqueued ahead of its start point (lsa_icp_gate / lsa_icp_link) -- x0 comes from there, or nothing is done
  // What this solve leaves for the ICP iteration enqueued behind it (lsa_icp_link; Slam.cxx:907, 940-950 and 1086, 1134-1151):
  // whether it runs at all, the pose its keypoints are searched under, its start point and -- localization -- the undistortion.
  IcpGate* leave;             // null: nothing is left (no iteration behind this one)
  int link_refine;            // localization with Slam::RefineUndistortion between two iterations
  int motion_from_args;       // the motion within the frame as the loop starts with it: motion0 (first solve of the loop) or motion_dev
  posemath::ScanPoseClock clock;
  Rigid previous_world;       // PreviousTworld
  double motion0[16];         // Time0, Time1, Rot0 (w x y z), Rot1, Trans0, Trans1
  double* motion_dev;         // the same 16 values between the solves of one loop
};

// result layout (doubles): [0..5] pose, [6] initial cost, [7] final cost, [8..36] the 29 sums at the final
// point, [37] successful steps, [38] unsuccessful steps, [39] iterations, [40] evaluations, [41] skipped,
// [42] termination code, [43] matches, [44] failure (a spin ran out)
constexpr int kResPose = 0, kResInitial = 6, kResFinal = 7, kResSums = 8, kResSuccessful = 37, kResUnsuccessful = 38, kResIterations = 39,
              kResEvaluations = 40, kResSkipped = 41, kResCode = 42, kResMatches = 43, kResFailed = 44, kResCount = 45;
static_assert(kResCount <= kLmOut, "result does not fit the mailbox");

// index of H(a, b), a <= b, in the 29 sums (cost, g[6], upper triangle row by row, count)
__device__ __forceinline__ constexpr int hidx(int a, int b) { return 7 + a * 6 - (a * (a - 1)) / 2 + (b - a); }
__device__ __forceinline__ constexpr int hsym(int a, int b) { return a <= b ? hidx(a, b) : hidx(b, a); }

// trust-region state between two evaluations (thread 0 of every block keeps its own copy in LDS: nothing of it is
// live in registers while the residual blocks are evaluated)
struct LmState
{
  int cur_buf;             // which of Shared::sums holds the sums at the current point x (the other one takes the next evaluation's)
  double x[6], scale[6], diag[6];
  double radius, decrease_factor, x_norm, model_cost_change, delta_norm, initial_cost;
  int reuse_diagonal, consecutive_invalid, iter;
  int evaluations, successful, unsuccessful, iterations, code, skipped, matches;
};

struct Shared
{
  double wsum[kLmThreads / 64][kAccumVals];
  unsigned gat[kLmBlocksMax][2 * kAccumVals];  // halves of every block's partial sums
  double part[kAccumVals][8];
  // The sums over ALL residual blocks, twice: at the current point, and of the evaluation under way / just done.  A step that
  // is accepted makes the second the first by flipping LmState::cur_buf (29 values that one thread no longer copies).
  double sums[2][kAccumVals];
  double Hs[36], gs[6];     // the scaled system of the step under way (lm_scale_system: a lane per entry)
  double w[6];              // the point to evaluate next
  double rot[39];           // R, dR/drx, dR/dry, dR/drz, t at w
  LmState lm;
  int failed;
  int stop;
  unsigned long long lap[6], tk;  // diagnostics (block 0): evaluate, exchange, fold, step [100 MHz ticks], evaluations, total
  unsigned long long sub[4], st0;  // diagnostics: parts of the step -- decision, scaled system, Cholesky solve, candidate
};

__device__ __forceinline__ double uniform(double v)
{
  const long long b = __double_as_longlong(v);
  const int lo = __builtin_amdgcn_readfirstlane((int)(b & 0xffffffffll)), hi = __builtin_amdgcn_readfirstlane((int)(b >> 32));
  return __longlong_as_double(((long long)hi << 32) | (long long)(unsigned)lo);
}

// One evaluation at sh.w / sh.rot: sh.sums[1 - cur_buf][] = the 29 sums over ALL residual blocks, identical in every block.
// Returns false when a spin ran out (uniform over the block).
__device__ __forceinline__ void lm_lap(Shared& sh, int slot)
{
  if (threadIdx.x == 0)
  {
    const unsigned long long now = wall_clock64();
    sh.lap[slot] += now - sh.tk;
    sh.tk = now;
  }
}

__device__ __forceinline__ bool lm_evaluate(const LmParams& p, unsigned epoch, u64* __restrict__ xchg, Shared& sh, double* __restrict__ cache, bool trace)
{
  {
    RotConst c;
    // the same values in every lane: kept in scalar registers
#pragma unroll
    for (int i = 0; i < 9; ++i) { c.R[i] = uniform(sh.rot[i]); c.dRx[i] = uniform(sh.rot[9 + i]); c.dRy[i] = uniform(sh.rot[18 + i]); c.dRz[i] = uniform(sh.rot[27 + i]); }
#pragma unroll
    for (int i = 0; i < 3; ++i) c.t[i] = uniform(sh.rot[36 + i]);
    double acc[kAccumVals];
#pragma unroll
    for (int v = 0; v < kAccumVals; ++v) acc[v] = 0.;
    const unsigned long long e0 = trace ? wall_clock64() : 0ull;
    accumulate_records_cached(p.set, c, blockIdx.x * blockDim.x + threadIdx.x, gridDim.x * blockDim.x, cache, p.cslots, p.cslots * kLmThreads, epoch == 1, kRecordPrefetch && p.cslots > 0 ? 1 : 0, acc);
    const unsigned long long e1 = trace ? wall_clock64() : 0ull;
    int slot;
    const double total = wave_reduce_accum(acc, slot);  // this lane's one value of the 29, summed over the wavefront
    if (trace && threadIdx.x == 0) { sh.lap[4] += e1 - e0; }
    if ((threadIdx.x & 1) == 0 && slot < kAccumVals) sh.wsum[threadIdx.x >> 6][slot] = total;
  }
  if (threadIdx.x == 0) sh.failed = 0;
  __syncthreads();
  if (trace) lm_lap(sh, 0);
  const unsigned tag = p.tag_base + epoch;
  const int nb = gridDim.x;
  u64* slots = xchg + (size_t)(epoch & 1u) * kLmBlocksMax * kMailboxStride;
  if (threadIdx.x < 2 * kAccumVals)
  {
    const int v = threadIdx.x >> 1, half = threadIdx.x & 1;
    double r = sh.wsum[0][v];
#pragma unroll
    for (int w = 1; w < kLmThreads / 64; ++w) r += sh.wsum[w][v];  // ((w0 + w1) + w2) + ...
    const u64 bits = (u64)__double_as_longlong(r);
    const unsigned word = half ? (unsigned)(bits >> 32) : (unsigned)(bits & 0xffffffffull);
    __hip_atomic_store(slots + (size_t)blockIdx.x * kMailboxStride + threadIdx.x, ((u64)tag << 32) | word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  // gather: every thread sweeps its granules, a batch of loads in flight together, until every tag matches
  const int total = nb * 2 * kAccumVals;
  bool failed = (int)blockIdx.x == p.give_up_block;  // (test hook: as if this workgroup had waited too long)
  {
    const unsigned long long t0 = wall_clock64();
    unsigned spins = 0;
    for (int base = threadIdx.x; base < total && !failed; base += 8 * kLmThreads)
    {
      while (true)
      {
        bool ok = true;
        u64 x[8];
#pragma unroll
        for (int k = 0; k < 8; ++k)
        {
          const int g = base + k * kLmThreads;
          if (g < total)
          {
            const int b = g / (2 * kAccumVals), s = g - b * (2 * kAccumVals);
            x[k] = __hip_atomic_load(slots + (size_t)b * kMailboxStride + s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
        }
#pragma unroll
        for (int k = 0; k < 8; ++k)
        {
          const int g = base + k * kLmThreads;
          if (g < total)
          {
            if ((unsigned)(x[k] >> 32) == tag)
            {
              const int b = g / (2 * kAccumVals), s = g - b * (2 * kAccumVals);
              sh.gat[b][s] = (unsigned)(x[k] & 0xffffffffull);
            }
            else ok = false;
          }
        }
        if (ok) break;
        // 100 MHz clock: 20 ms without the other blocks' sums (they are not resident, or gave up themselves)
        if ((++spins & 15u) == 0 && wall_clock64() - t0 > 2000000ull) { failed = true; break; }
        __builtin_amdgcn_s_sleep(2);
      }
      if (failed) break;
    }
  }
  if (failed) atomicOr(&sh.failed, 1);
  __syncthreads();
  if (trace) lm_lap(sh, 1);
  if (sh.failed) return false;
  // fold in a fixed order: 8 partial sums per value over the blocks b = j, j + 8, ..., then the 8 in order
  if (threadIdx.x < kAccumVals * 8)
  {
    const int v = threadIdx.x >> 3, j = threadIdx.x & 7;
    double s = 0.;
    for (int b = j; b < nb; b += 8)
    {
      const u64 bits = ((u64)sh.gat[b][2 * v + 1] << 32) | sh.gat[b][2 * v];
      s += __longlong_as_double((long long)bits);
    }
    sh.part[v][j] = s;
  }
  __syncthreads();
  if (threadIdx.x < kAccumVals)
  {
    double s = sh.part[threadIdx.x][0];
#pragma unroll
    for (int j = 1; j < 8; ++j) s += sh.part[threadIdx.x][j];
    sh.sums[1 - sh.lm.cur_buf][threadIdx.x] = s;
  }
  __syncthreads();
  if (trace) lm_lap(sh, 2);
  return true;
}

// dense symmetric positive definite solve, as SolveSPD of host/lsa_lm.cpp: Cholesky, forward, backward
template <int N>
__device__ __forceinline__ bool solve_spd(const double A[N * N], const double b[N], double x[N])
{
  // (every division by a diagonal element is a multiplication by its reciprocal, taken once: 6 divisions instead of 27
  // on the one thread that runs the step; host/lsa_lm.cpp and the oracle do the same, operation for operation)
  double L[N * N], rinv[N];
#pragma unroll
  for (int i = 0; i < N * N; ++i) L[i] = 0.;
  bool ok = true;
#pragma unroll
  for (int i = 0; i < N; ++i)
#pragma unroll
    for (int j = 0; j <= i; ++j)
    {
      double s = A[i * N + j];
#pragma unroll
      for (int k = 0; k < j; ++k) s -= L[i * N + k] * L[j * N + k];
      if (i == j)
      {
        if (!(s > 0.0) || !isfinite(s)) ok = false;
        L[i * N + i] = __builtin_sqrt(s);
        rinv[i] = 1.0 / L[i * N + i];
      }
      else
        L[i * N + j] = s * rinv[j];
    }
  if (!ok) return false;
  double y[N];
#pragma unroll
  for (int i = 0; i < N; ++i)
  {
    double s = b[i];
#pragma unroll
    for (int k = 0; k < i; ++k) s -= L[i * N + k] * y[k];
    y[i] = s * rinv[i];
  }
#pragma unroll
  for (int i = N - 1; i >= 0; --i)
  {
    double s = y[i];
#pragma unroll
    for (int k = i + 1; k < N; ++k) s -= L[k * N + i] * x[k];
    x[i] = s * rinv[i];
  }
#pragma unroll
  for (int i = 0; i < N; ++i)
    if (!isfinite(x[i])) ok = false;
  return ok;
}

enum LmCode
{
  kCodeNone = 0, kCodeNotEnoughMatches, kCodeGradient0, kCodeMaxIterations, kCodeGradient, kCodeMinRadius, kCodeInvalidSteps, kCodeParameterTolerance,
  kCodeFunctionTolerance
};

// sh.w = the point to evaluate next (one thread); its rotation and derivatives follow in finish_point
__device__ __forceinline__ void set_point(Shared& sh, const double w[6])
{
#pragma unroll
  for (int a = 0; a < 6; ++a) sh.w[a] = w[a];
}
__device__ __forceinline__ double lane_value(double v, int lane)
{
  const long long b = __double_as_longlong(v);
  const int lo = __shfl((int)(b & 0xffffffffll), lane), hi = __shfl((int)(b >> 32), lane);
  return __longlong_as_double(((long long)hi << 32) | (long long)(unsigned)lo);
}
// sh.rot = rotation and derivatives at sh.w (CeresCostFunctions.h:67-79).  The first wavefront runs it, all lanes: the
// three cosines and three sines -- a few hundred dependent instructions each, the longest part of what lies between two
// evaluations -- are worked out by six lanes side by side instead of one lane six times.
__device__ __forceinline__ void finish_point(Shared& sh)
{
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // sh.w was written by lane 0 of this wavefront
  const int lane = threadIdx.x & 63;
  const double angle = sh.w[3 + (lane < 6 ? lane % 3 : 0)];
  const double v = lane < 3 ? lsa_cos(angle) : lsa_sin(angle);
  const double cx = lane_value(v, 0), cy = lane_value(v, 1), cz = lane_value(v, 2);
  const double sx = lane_value(v, 3), sy = lane_value(v, 4), sz = lane_value(v, 5);
  if (lane == 0)
  {
    double R[9], dRx[9], dRy[9], dRz[9];
    rotation_and_derivatives(cx, sx, cy, sy, cz, sz, R, dRx, dRy, dRz);
#pragma unroll
    for (int i = 0; i < 9; ++i) { sh.rot[i] = R[i]; sh.rot[9 + i] = dRx[i]; sh.rot[18 + i] = dRy[i]; sh.rot[27 + i] = dRz[i]; }
#pragma unroll
    for (int i = 0; i < 3; ++i) sh.rot[36 + i] = sh.w[i];
  }
}

// What the trust-region loop of host/lsa_lm.cpp (LocalOptimizer::Solve) does between two evaluations, N active
// parameters: all 6, or (x, y, rz) in 2D mode (SubsetParameterization(6, {2, 3, 4}), LocalOptimizer.cxx:89-90).
// In: sh.sums[1 - cur_buf] = the sums at sh.w (the start point when first, a candidate afterwards).  Out: sh.w / sh.rot =
// the next candidate, or true = the solve is over.  Three parts, the first wavefront runs them: lm_decide (one lane: what
// became of the candidate), lm_scale_system (a lane per entry of the Jacobi-scaled normal equations), lm_take_step (one lane:
// the damped solve and the next candidate).  The arithmetic of every value is the host loop's, operation for operation.
template <int N> __device__ __forceinline__ constexpr int lm_act(int a) { return N == 6 ? a : (a == 2 ? 5 : a); }
template <int N>
__device__ __forceinline__ double lm_grad_max(const double* cur)
{
  double m = 0;
#pragma unroll
  for (int a = 0; a < N; ++a) { const double v = __builtin_fabs(cur[1 + lm_act<N>(a)]); m = m > v ? m : v; }  // std::max(m, v)
  return m;
}
template <int N>
__device__ bool lm_decide(const LmParams& p, Shared& sh, bool first)
{
  const double function_tolerance = 1e-6, gradient_tolerance = 1e-10, parameter_tolerance = 1e-8;
  const double min_relative_decrease = 1e-3, max_radius = 1e16;
  LmState& lm = sh.lm;
  const double* cur = sh.sums[lm.cur_buf];
  const double* tot = sh.sums[1 - lm.cur_buf];
  auto xnorm = [&]() {
    double s = 0;
#pragma unroll
    for (int a = 0; a < N; ++a) s += lm.x[lm_act<N>(a)] * lm.x[lm_act<N>(a)];
    return __builtin_sqrt(s);
  };

  if (first)
  {
    lm.matches = (int)tot[28];
    if (lm.matches < p.min_matches)
    {
      lm.skipped = 1;
      lm.code = kCodeNotEnoughMatches;
      return true;
    }
    lm.cur_buf ^= 1;  // the sums at the start point are the current ones
    cur = tot;
    lm.initial_cost = cur[0];
    lm.successful = 1;  // iteration 0 is reported as a successful step by Ceres
#pragma unroll
    for (int a = 0; a < N; ++a) lm.scale[a] = 1.0 / (1.0 + __builtin_sqrt(cur[hidx(lm_act<N>(a), lm_act<N>(a))]));  // Jacobi scaling, once
    if (lm_grad_max<N>(cur) <= gradient_tolerance) { lm.code = kCodeGradient0; return true; }
    lm.x_norm = xnorm();
  }
  else
  {
    // parameter / function tolerance terminate WITHOUT taking the candidate step
    if (lm.delta_norm <= parameter_tolerance * (lm.x_norm + parameter_tolerance)) { lm.code = kCodeParameterTolerance; return true; }
    const double cost_change = cur[0] - tot[0];
    if (__builtin_fabs(cost_change) <= function_tolerance * cur[0]) { lm.code = kCodeFunctionTolerance; return true; }
    const double relative_decrease = cost_change / lm.model_cost_change;
    if (relative_decrease > min_relative_decrease)
    {
#pragma unroll
      for (int a = 0; a < 6; ++a) lm.x[a] = sh.w[a];
      lm.x_norm = xnorm();
      // cost AND Jacobian were evaluated at the candidate (Ceres evaluates the Jacobian again on acceptance, at the
      // same point: same arithmetic, one evaluation less): its sums become the current ones
      lm.cur_buf ^= 1;
      ++lm.successful;
      const double t = 2.0 * relative_decrease - 1.0;
      const double q = 1.0 - t * t * t;
      const double d = (1.0 / 3.0) < q ? q : (1.0 / 3.0);  // std::max(1.0 / 3.0, q)
      const double r = lm.radius / d;
      lm.radius = r < max_radius ? r : max_radius;          // std::min(max_radius, r)
      lm.decrease_factor = 2.0;
      lm.reuse_diagonal = 0;
    }
    else
    {
      ++lm.unsuccessful;
      lm.radius /= lm.decrease_factor; lm.decrease_factor *= 2.0; lm.reuse_diagonal = 1;
    }
  }
  if (sh.st0) { const unsigned long long n = wall_clock64(); sh.sub[0] += n - sh.st0; sh.st0 = n; }
  return false;
}
// sh.Hs / sh.gs = the normal equations at the current point, scaled: lane l < N * N the entry l, the next N lanes the gradient
template <int N>
__device__ __forceinline__ void lm_scale_system(Shared& sh)
{
  const int l = threadIdx.x & 63;
  const LmState& lm = sh.lm;
  const double* cur = sh.sums[lm.cur_buf];
  if (l < N * N)
  {
    const int a = l / N, b = l - a * N;
    const int fa = lm_act<N>(a), fb = lm_act<N>(b);
    const int lo = fa <= fb ? fa : fb, hi = fa <= fb ? fb : fa;
    sh.Hs[l] = cur[7 + lo * 6 - (lo * (lo - 1)) / 2 + (hi - lo)] * lm.scale[a] * lm.scale[b];  // hsym(fa, fb)
  }
  else if (l < N * N + N)
  {
    const int a = l - N * N;
    sh.gs[a] = cur[1 + lm_act<N>(a)] * lm.scale[a];
  }
}
template <int N>
__device__ bool lm_take_step(const LmParams& p, Shared& sh)
{
  const double gradient_tolerance = 1e-10, min_trust_region_radius = 1e-32;
  const double min_diagonal = 1e-6, max_diagonal = 1e32;
  const int max_consecutive_invalid = 5;
  LmState& lm = sh.lm;
  const double* cur = sh.sums[lm.cur_buf];
  double Hs[N * N], gs[N];
#pragma unroll
  for (int i = 0; i < N * N; ++i) Hs[i] = sh.Hs[i];
#pragma unroll
  for (int a = 0; a < N; ++a) gs[a] = sh.gs[a];
  while (true)
  {
    if (lm.iter >= p.max_iter) { lm.code = kCodeMaxIterations; return true; }
    if (lm_grad_max<N>(cur) <= gradient_tolerance) { lm.code = kCodeGradient; return true; }
    if (lm.radius < min_trust_region_radius) { lm.code = kCodeMinRadius; return true; }
    ++lm.iter;
    lm.iterations = lm.iter;
    if (!lm.reuse_diagonal)
    {
#pragma unroll
      for (int a = 0; a < N; ++a)
      {
        const double h = Hs[a * N + a];
        const double lo = h < min_diagonal ? min_diagonal : h;  // std::max(h, min_diagonal)
        lm.diag[a] = max_diagonal < lo ? max_diagonal : lo;     // std::min(lo, max_diagonal)
      }
    }
    double M[N * N], y[N], step[N];
#pragma unroll
    for (int i = 0; i < N * N; ++i) M[i] = Hs[i];
#pragma unroll
    for (int a = 0; a < N; ++a) M[a * N + a] += lm.diag[a] / lm.radius;
#pragma unroll
    for (int a = 0; a < N; ++a) y[a] = 0.;
    if (sh.st0) { const unsigned long long n = wall_clock64(); sh.sub[1] += n - sh.st0; sh.st0 = n; }
    bool ok = solve_spd<N>(M, gs, y);
    if (sh.st0) { const unsigned long long n = wall_clock64(); sh.sub[2] += n - sh.st0; sh.st0 = n; }
    lm.reuse_diagonal = 1;
    double model_cost_change = 0;
    if (ok)
    {
#pragma unroll
      for (int a = 0; a < N; ++a) step[a] = -y[a];
      double sg = 0, sHs = 0;
#pragma unroll
      for (int a = 0; a < N; ++a)
      {
        sg += step[a] * gs[a];
        double t = 0;
#pragma unroll
        for (int b = 0; b < N; ++b) t += Hs[a * N + b] * step[b];
        sHs += step[a] * t;
      }
      model_cost_change = -sg - 0.5 * sHs;
      if (model_cost_change < 0.0) ok = false;
    }
    if (!ok)
    {
      ++lm.unsuccessful;
      if (++lm.consecutive_invalid >= max_consecutive_invalid) { lm.code = kCodeInvalidSteps; return true; }
      lm.radius /= lm.decrease_factor; lm.decrease_factor *= 2.0; lm.reuse_diagonal = 1;
      continue;
    }
    lm.consecutive_invalid = 0;
    lm.model_cost_change = model_cost_change;

    double cand[6], delta_norm = 0;
#pragma unroll
    for (int a = 0; a < 6; ++a) cand[a] = lm.x[a];
#pragma unroll
    for (int a = 0; a < N; ++a)
    {
      cand[lm_act<N>(a)] = lm.x[lm_act<N>(a)] + step[a] * lm.scale[a];
      const double e = lm.x[lm_act<N>(a)] - cand[lm_act<N>(a)];
      delta_norm += e * e;
    }
    lm.delta_norm = __builtin_sqrt(delta_norm);
    set_point(sh, cand);
    ++lm.evaluations;
    if (sh.st0) { const unsigned long long n = wall_clock64(); sh.sub[3] += n - sh.st0; sh.st0 = n; }
    return false;
  }
}
// the three parts, by the first wavefront (all its lanes call); sh.stop = the solve is over
template <int N>
__device__ __forceinline__ void lm_step(const LmParams& p, Shared& sh, bool first)
{
  if (threadIdx.x == 0) sh.stop = lm_decide<N>(p, sh, first) ? 1 : 0;
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // one wavefront: its LDS operations are served in order
  if (sh.stop) return;
  lm_scale_system<N>(sh);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  if (threadIdx.x == 0) sh.stop = lm_take_step<N>(p, sh) ? 1 : 0;
}

// What the host's loop does between two ICP iterations (host/lsa_slam_core.cpp, Slam.cxx:940-950 / 1134-1151), for the
// iteration enqueued behind this solve: does it run (the solve was not skipped and made a step), the pose from the solve's
// parameters, the next start point (the round trip through the matrix, as LocalOptimizer::SetPosePrior makes it) and,
// localization, Slam::RefineUndistortion under the new pose.  lsa_posemath.h is the host's arithmetic: the host works
// the same values out from the same result when it arrives, bit for bit.  One wavefront: the six half-angle
// cosines / sines side by side, the rest on lane 0.
__device__ void leave_link(const LmParams& p, const LmState& lm, bool failed, int part)
{
  using namespace posemath;
  IcpGate* g = p.leave;
  const int lane = threadIdx.x & 63;
  const bool go = !failed && !lm.skipped && lm.successful != 1;
  if (!go)
  {
    if (lane == 0 && part == 0) g->go = 0ull;
    return;
  }
  // two wavefronts side by side (the block is read after the kernel has ended: no order among its words): part 0 -- go,
  // pose, start point; part 1 -- the undistortion.  Both work the pose out for themselves (the same bits).
  if (part == 1 && !p.link_refine) return;
  const double half = lm.x[3 + (lane < 6 ? lane % 3 : 0)] * 0.5;
  const double v = lane < 3 ? lsa_cos(half) : lsa_sin(half);
  const double cx = lane_value(v, 0), cy = lane_value(v, 1), cz = lane_value(v, 2);
  const double sx = lane_value(v, 3), sy = lane_value(v, 4), sz = lane_value(v, 5);
  double x[6];
#pragma unroll
  for (int a = 0; a < 6; ++a) x[a] = lm.x[a];
  const Pose T = FromXYZRPYTrig(x, cx, sx, cy, sy, cz, sz);
  if (part == 0)
  {
    if (lane != 0) return;
    double x0[6];
    ToXYZRPY(T, x0);
    g->in.pose = ToRigid(T);
#pragma unroll
    for (int a = 0; a < 6; ++a) g->in.x0[a] = x0[a];
    g->go = 1ull;
    return;
  }
  WithinFrameMotion m;
  double mv[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) mv[i] = p.motion0[i];
  if (!p.motion_from_args)
  {
#pragma unroll
    for (int i = 0; i < 16; ++i) mv[i] = p.motion_dev[i];
  }
  m.Time0 = mv[0]; m.Time1 = mv[1];
  m.Rot0 = {mv[2], mv[3], mv[4], mv[5]};
  m.Rot1 = {mv[6], mv[7], mv[8], mv[9]};
#pragma unroll
  for (int i = 0; i < 3; ++i) { m.Trans0[i] = mv[10 + i]; m.Trans1[i] = mv[13 + i]; }
  // Slam::RefineUndistortion with the scan's pose at its begin (lane 0) and at its end (lane 1) interpolated side by
  // side; lane 0 takes the end over and goes on alone
  const Pose previous = FromRigid(p.previous_world);
  const Pose mine = InterpolateScanPose(p.clock, previous, T, lane == 1 ? m.Time1 : m.Time0);
  Pose worldToBaseEnd = Pose::Identity();
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) worldToBaseEnd(i, j) = lane_value(mine(i, j), 1);
  if (lane != 0) return;
  Pose d0, d1;
  RefineUndistortionFrom(m, mine, worldToBaseEnd, T, d0, d1);
  g->in.ic = MakeInterpConst(d0, d1, m.Time0, m.Time1);
  double* o = p.motion_dev;
  o[0] = m.Time0; o[1] = m.Time1;
  o[2] = m.Rot0.w; o[3] = m.Rot0.x; o[4] = m.Rot0.y; o[5] = m.Rot0.z;
  o[6] = m.Rot1.w; o[7] = m.Rot1.x; o[8] = m.Rot1.y; o[9] = m.Rot1.z;
#pragma unroll
  for (int i = 0; i < 3; ++i) { o[10 + i] = m.Trans0[i]; o[13 + i] = m.Trans1[i]; }
}

__global__ __launch_bounds__(kLmThreads) void k_lm_solve(LmParams p, u64* __restrict__ xchg, u64* __restrict__ mailbox, unsigned out_tag, unsigned long long* trace)
{
  __shared__ Shared sh;
  extern __shared__ double lm_cache[];  // [17][cslots * kLmThreads]
  const bool tr = trace != nullptr && blockIdx.x == 0;
  // the thread's first residual block: on its way from memory while the launch finds out whether it runs and where it starts
  RecordRegs pre;
  if (kRecordPrefetch && p.cslots > 0) record_load(p.set, blockIdx.x * blockDim.x + threadIdx.x, pre);
  if (p.gate)
  {
    // enqueued ahead of its start point: the gate in front of this launch has left it (go == 1), or the iteration was
    // called off (0: nobody waits for a result) or the gate gave up waiting for the host (2: the host is told)
    const unsigned long long go = p.gate->go;
    if (go != 1ull)
    {
      if (go == 2ull && blockIdx.x == 0 && threadIdx.x < 2 * kResCount)
      {
        const double r = (threadIdx.x >> 1) == kResFailed ? 2. : 0.;
        const u64 bits = (u64)__double_as_longlong(r);
        const unsigned word = (threadIdx.x & 1) ? (unsigned)(bits >> 32) : (unsigned)(bits & 0xffffffffull);
        __hip_atomic_store(mailbox + threadIdx.x, ((u64)out_tag << 32) | word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      }
      // an iteration that does not run leaves the same for the one behind it
      if (p.leave && blockIdx.x == 0 && threadIdx.x == 0) p.leave->go = 0ull;
      return;
    }
  }
  if (threadIdx.x == 0)
  {
    for (int i = 0; i < 6; ++i) sh.lap[i] = 0;
    for (int i = 0; i < 4; ++i) sh.sub[i] = 0;
    sh.st0 = 0;
    sh.tk = wall_clock64();
    sh.lap[5] = sh.tk;
    LmState& lm = sh.lm;
#pragma unroll
    for (int v = 0; v < kAccumVals; ++v) sh.sums[0][v] = 0.;
    lm.cur_buf = 0;
    // the start point: the launch's own argument, or what the gate brought over (read into LDS, the arguments stay untouched)
    double x0[6];
#pragma unroll
    for (int a = 0; a < 6; ++a) x0[a] = p.x0[a];
    if (p.gate)
    {
#pragma unroll
      for (int a = 0; a < 6; ++a) x0[a] = p.gate->in.x0[a];
    }
#pragma unroll
    for (int a = 0; a < 6; ++a) { lm.x[a] = x0[a]; lm.scale[a] = 1.; lm.diag[a] = 0.; }
    lm.radius = 1e4; lm.decrease_factor = 2.0; lm.x_norm = 0.; lm.model_cost_change = 0.; lm.delta_norm = 0.; lm.initial_cost = 0.;
    lm.reuse_diagonal = 0; lm.consecutive_invalid = 0; lm.iter = 0;
    lm.evaluations = 1; lm.successful = 0; lm.unsuccessful = 0; lm.iterations = 0; lm.code = kCodeNone; lm.skipped = 0; lm.matches = 0;
    set_point(sh, x0);
    sh.stop = 0;
  }
  if (threadIdx.x < 64) finish_point(sh);
  if (kRecordPrefetch && p.cslots > 0) record_stash(pre, lm_cache, p.cslots * kLmThreads, threadIdx.x);
  __syncthreads();
  bool failed = false;
  // every evaluation of the launch has an epoch of its own; all blocks walk through the same sequence of them
  for (unsigned epoch = 1;; ++epoch)
  {
    if (!lm_evaluate(p, epoch, xchg, sh, lm_cache, tr)) { failed = true; break; }
    if (threadIdx.x < 64)
    {
      if (threadIdx.x == 0 && tr) sh.st0 = wall_clock64();
      if (p.two_d) lm_step<3>(p, sh, epoch == 1);
      else lm_step<6>(p, sh, epoch == 1);
      finish_point(sh);
    }
    __syncthreads();
    if (tr) lm_lap(sh, 3);
    if (sh.stop) break;
  }
  if (blockIdx.x != 0) return;
  if (tr && threadIdx.x == 0)
  {
    for (int i = 0; i < 4; ++i) trace[i] += sh.lap[i];
    trace[4] += (unsigned long long)sh.lm.evaluations;
    trace[5] += wall_clock64() - sh.lap[5];
    trace[6] += 1;
    trace[7] += sh.lap[4];  // of "evaluate": the residual blocks alone (before the wavefront's reduction)
    for (int i = 0; i < 4; ++i) trace[8 + i] += sh.sub[i];
  }
  // the iteration enqueued behind this solve: the second and third wavefronts prepare what it reads while the first sends the result
  if (p.leave && threadIdx.x >= 64 && threadIdx.x < 192) leave_link(p, sh.lm, failed, (threadIdx.x >> 6) - 1);
  // every block holds the same result; block 0's copy goes out as 2 granules per double
  if (threadIdx.x < 2 * kResCount)
  {
    const LmState& lm = sh.lm;
    const int v = threadIdx.x >> 1, half = threadIdx.x & 1;
    double r = 0.;
    if (v < kResInitial) r = lm.x[v - kResPose];
    else if (v == kResInitial) r = lm.initial_cost;
    else if (v == kResFinal) r = sh.sums[lm.cur_buf][0];
    else if (v < kResSums + kAccumVals) r = sh.sums[lm.cur_buf][v - kResSums];
    else if (v == kResSuccessful) r = lm.successful;
    else if (v == kResUnsuccessful) r = lm.unsuccessful;
    else if (v == kResIterations) r = lm.iterations;
    else if (v == kResEvaluations) r = lm.evaluations;
    else if (v == kResSkipped) r = lm.skipped;
    else if (v == kResCode) r = lm.code;
    else if (v == kResMatches) r = lm.matches;
    else if (v == kResFailed) r = failed ? 1. : 0.;
    const u64 bits = (u64)__double_as_longlong(r);
    const unsigned word = half ? (unsigned)(bits >> 32) : (unsigned)(bits & 0xffffffffull);
    __hip_atomic_store(mailbox + threadIdx.x, ((u64)out_tag << 32) | word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

// Waits for the host to release gate `seq`, then leaves what the host posted in the device block the launches behind the
// gate read.  The host's block is 128 granules {seq, half of word i / 2}; the two wavefronts sweep all of them with one
// load each, again and again, until every tag is `seq`: one trip over the bus after the host's last store, and no
// assumption on the order the reads are served in.  Word 0 of the block is go (1 run, 0 called off); the wait is bounded
// (50 ms of the 100 MHz clock): a host that does not answer makes the launches behind it do nothing and say so (go = 2).
__global__ __launch_bounds__(kGateGranules) void k_icp_gate(const u64* __restrict__ host_granules, u64* __restrict__ dev_words, unsigned seq, int give_up)
{
  __shared__ unsigned halves[kGateGranules];
  __shared__ int state;  // 0 waiting, 1 all there, 2 gave up
  const int t = threadIdx.x;
  if (t == 0) state = give_up ? 2 : 0;  // (test hook: as if the host had not answered in time)
  __syncthreads();
  const unsigned long long t0 = wall_clock64();
  while (state == 0)
  {
    const u64 g = __hip_atomic_load(host_granules + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    const bool mine = (unsigned)(g >> 32) == seq;
    if (mine) halves[t] = (unsigned)(g & 0xffffffffull);
    const int all = __syncthreads_and(mine ? 1 : 0);
    if (t == 0)
    {
      if (all) state = 1;
      else if (wall_clock64() - t0 > 5000000ull) state = 2;
    }
    __syncthreads();
    if (state == 0) __builtin_amdgcn_s_sleep(1);
  }
  if (state == 1)
  {
    if (t < kGateWords)
    {
      const u64 w = ((u64)halves[2 * t + 1] << 32) | halves[2 * t];
      // called off (go == 0): only the first word matters
      if (t == 0 || (halves[0] | halves[1]) != 0u) dev_words[t] = w;
    }
  }
  else if (t == 0)
    dev_words[0] = 2ull;
}

const char* const kMessages[] = {"", "not enough matches", "gradient tolerance (iteration 0)", "max iterations", "gradient tolerance",
                                 "min trust region radius", "too many invalid steps", "parameter tolerance", "function tolerance"};

}  // namespace

namespace lsa
{
std::atomic<int> g_live_contexts{0};
// workgroups one solve may use so that the solves that can run at once all fit on the chip
int lm_blocks_share()
{
  static const int cus = [] {
    hipDeviceProp_t prop;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 256;
    return std::max(prop.multiProcessorCount, 8);
  }();
  static const int queues = [] {
    const char* e = std::getenv("GPU_MAX_HW_QUEUES");
    const int q = e ? std::atoi(e) : 4;  // the runtime's default
    return std::max(q, 1);
  }();
  const int side_by_side = std::max(1, std::min(g_live_contexts.load(std::memory_order_relaxed), queues));
  return std::max(cus / side_by_side, 8);
}
InterpConst make_interp_const(const double H0[16], const double H1[16], double t0, double t1);  // lsa_transform.hip
// how many 256-thread layers of residual blocks (34 KB each) the solve kernel may keep in LDS beside its own data
int lm_cache_capacity()
{
  int slots = 0;
  for (int want = 3; want >= 1 && slots == 0; --want)
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(k_lm_solve), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(want * 17 * kLmThreads * sizeof(double))) == hipSuccess)
      slots = want;
  (void)hipGetLastError();
  return slots;
}
}  // namespace lsa

extern "C" {

static int solve_device_begin(lsa_ctx* ctx, unsigned type_mask, const double prior[6], int two_d_mode, int lm_max_iter, int min_matches, int leave_ticket, const lsa_icp_link_t* link)
{
  if (!ctx) return LSA_E_ARG;
  if (leave_ticket >= 0 && (leave_ticket >= kGateRing || !link || !ctx->gate_dev || !ctx->gate_saved[leave_ticket].used || !ctx->gate_saved[leave_ticket].link))
    return ctx->fail(LSA_E_ARG, "lsa_solve_device_begin_linked: no such link (lsa_icp_link)");
  if (!prior && ctx->gate_current < 0) return ctx->fail(LSA_E_ARG, "lsa_solve_device_begin: no start point and no gate to wait behind");
  if (!ctx->lm_mailbox) return ctx->fail(LSA_E_STATE, "lsa_solve_device: no coherent host memory for the result");
  LSA_HIP(ctx, hipSetDevice(ctx->device));
  LmParams p;
  int total = 0;
  for (int k = 0; k < 3; ++k)
  {
    MatchBuf& mb = ctx->match[k];
    const bool use = (type_mask >> k) & 1u && mb.valid && mb.k > 0;
    p.set.rec[k] = mb.rec; p.set.status[k] = mb.status; p.set.cap[k] = mb.cap;
    p.set.count[k] = use ? mb.k : 0;
    p.set.sat2[k] = mb.sat * mb.sat;
    total += p.set.count[k];
  }
  for (int a = 0; a < 6; ++a) p.x0[a] = prior ? prior[a] : 0.;
  p.give_up_block = ctx->debug_lm_give_up_block;
  ctx->debug_lm_give_up_block = -1;
  p.gate = prior ? nullptr : reinterpret_cast<const IcpGate*>(ctx->gate_dev + (size_t)ctx->gate_current * kGateWords);
  p.leave = nullptr;
  p.link_refine = 0;
  p.motion_from_args = 0;
  p.motion_dev = ctx->motion_dev;
  std::memset(&p.clock, 0, sizeof(p.clock));
  std::memset(&p.previous_world, 0, sizeof(p.previous_world));
  std::memset(p.motion0, 0, sizeof(p.motion0));
  if (leave_ticket >= 0)
  {
    p.leave = reinterpret_cast<IcpGate*>(ctx->gate_dev + (size_t)leave_ticket * kGateWords);
    p.link_refine = link->refine_undistortion ? 1 : 0;
    p.motion_from_args = link->first ? 1 : 0;
    p.clock.have_log = link->have_log ? 1 : 0;
    p.clock.prev_time = link->prev_time;
    p.clock.cur_time = link->cur_time;
    p.clock.max_ratio = link->max_extrapolation_ratio;
    row_major_to_rt(link->previous_world, p.previous_world.R, p.previous_world.t);
    std::memcpy(p.motion0, link->motion, sizeof(p.motion0));
  }
  p.two_d = two_d_mode ? 1 : 0;
  p.max_iter = lm_max_iter < 0 ? 0 : lm_max_iter;
  p.min_matches = min_matches;
  // every evaluation of the launch takes a tag of its own; tags of earlier launches never come back within 2^32
  p.tag_base = ctx->lm_tag;
  ctx->lm_tag += (unsigned)p.max_iter + 4u;
  const unsigned out_tag = (unsigned)(++ctx->lm_seq);
  // about two residual blocks per thread (measured: 1024 per workgroup 63 us a solve, 512: 56, 256: 58 -- fewer blocks per thread
  // shorten the evaluation, more workgroups lengthen the exchange), never more workgroups than the exchange has slots for
  // ... and never more than this solve's share of the chip: the workgroups exchange their sums by waiting for each other,
  // so ALL of them have to be resident (512 threads of 250 registers: one workgroup per CU), also when the solves of
  // other contexts of this process run beside this one -- as many at once as the process has hardware queues
  const int nb = std::min(std::max((total + ctx->lm_records - 1) / ctx->lm_records, 1), std::min(std::min(ctx->lm_blocks, kLmBlocksMax), lm_blocks_share()));
  // the thread's first residual blocks stay in LDS between the evaluations (17 doubles each): as many per thread as
  // the block's share needs, as many as the LDS holds beside the kernel's own 34 KB (the rest is read again)
  const int per_thread = (total + nb * kLmThreads - 1) / (nb * kLmThreads);
  const size_t slot_bytes = (size_t)17 * kLmThreads * sizeof(double);
  p.cslots = std::min(per_thread, std::max(ctx->lm_cache_slots, 0));
  {
    ProfScope ps(ctx, "lm_solve", 0.);
    hipLaunchKernelGGL(k_lm_solve, dim3(nb), dim3(kLmThreads), p.cslots * slot_bytes, ctx->stream, p, ctx->lm_xchg, ctx->lm_mailbox + (size_t)(out_tag % kLmMailRing) * 2 * kLmOut, out_tag,
                       ctx->route_stats && ctx->trace_dev ? reinterpret_cast<unsigned long long*>(ctx->trace_dev) + (size_t)8192 * 12 : nullptr);
  }
  ctx->lm_pending.push_back(out_tag);
  ctx->lm_pending_wait.push_back(prior ? -1 : ctx->gate_current);
  // what is enqueued next waits behind the link this solve leaves
  if (leave_ticket >= 0) ctx->gate_current = leave_ticket;
  if (lsa_icp_trace_on()) std::fprintf(stderr, "[lm begin] tag %u gated %d leaves %d sat2 %.6g %.6g counts %d %d\n", out_tag, prior ? 0 : 1, leave_ticket, p.set.sat2[0], p.set.sat2[1], p.set.count[0], p.set.count[1]);
  return LSA_OK;
}

int lsa_solve_device_begin(lsa_ctx* ctx, unsigned type_mask, const double prior[6], int two_d_mode, int lm_max_iter, int min_matches)
{
  return solve_device_begin(ctx, type_mask, prior, two_d_mode, lm_max_iter, min_matches, -1, nullptr);
}

int lsa_solve_device_begin_linked(lsa_ctx* ctx, unsigned type_mask, const double prior[6], int two_d_mode, int lm_max_iter, int min_matches, int leave_ticket, const lsa_icp_link_t* link)
{
  if (ctx && (int)ctx->lm_pending.size() >= kLmMailRing - 1) return ctx->fail(LSA_E_STATE, "lsa_solve_device_begin_linked: too many solves in flight");
  return solve_device_begin(ctx, type_mask, prior, two_d_mode, lm_max_iter, min_matches, leave_ticket, link);
}

int lsa_solve_device_drop(lsa_ctx* ctx)
{
  if (!ctx || ctx->lm_pending.empty()) return LSA_E_ARG;
  ctx->lm_pending.pop_back();  // the solve begun last will never run (its gate was called off): nobody waits for it
  ctx->lm_pending_wait.pop_back();
  return LSA_OK;
}

int lsa_solve_device_end(lsa_ctx* ctx, lsa_solve_result_t* out)
{
  if (!ctx || !out) return ctx ? ctx->fail(LSA_E_ARG, "lsa_solve_device_end: bad argument") : LSA_E_ARG;
  if (ctx->lm_pending.empty()) return ctx->fail(LSA_E_STATE, "lsa_solve_device_end: no solve in flight");
  const unsigned out_tag = ctx->lm_pending.front();
  ctx->lm_pending.pop_front();
  const int waited_behind = ctx->lm_pending_wait.front();
  ctx->lm_pending_wait.pop_front();
  if (lsa_icp_trace_on()) std::fprintf(stderr, "[lm end] waits for tag %u (%zu more in flight)\n", out_tag, ctx->lm_pending.size());
  // the result arrives as granules in coherent host memory (as lsa_accumulate's sums do)
  double res[kResCount];
  const unsigned long long* box = ctx->lm_mailbox + (size_t)(out_tag % kLmMailRing) * 2 * kLmOut;  // (solves in flight do not share a mailbox)
  {
    const auto t0 = std::chrono::steady_clock::now();
    unsigned spins = 0;
    for (int v = 0; v < kResCount; ++v)
    {
      unsigned long long g[2];
      for (int h = 0; h < 2; ++h)
        while (((g[h] = __atomic_load_n(box + 2 * v + h, __ATOMIC_RELAXED)) >> 32) != out_tag)
        {
          if ((++spins & 0x3ff) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(2000))
          {
            // whatever was enqueued ahead is called off first: its gates wait for this thread (the stream would not
            // drain), and what its matches announced on the host -- the NEXT iteration's saturation distance -- must not
            // be what the caller's fall-back solves with
            (void)lsa_icp_abandon(ctx);
            LSA_HIP(ctx, hipStreamSynchronize(ctx->stream));
            return ctx->fail(LSA_E_STATE, "lsa_solve_device: no result from the device");
          }
#if defined(__x86_64__)
          __builtin_ia32_pause();
#endif
        }
      const unsigned long long bits = ((g[1] & 0xffffffffull) << 32) | (g[0] & 0xffffffffull);
      std::memcpy(&res[v], &bits, sizeof(double));
    }
  }
  if (lsa_icp_trace_on()) std::fprintf(stderr, "[lm end] tag %u: failed %g code %g evals %g matches %g\n", out_tag, res[kResFailed], res[kResCode], res[kResEvaluations], res[kResMatches]);
  // (the gate's case first and without waiting for the stream: the next iteration's gate may be waiting there for this thread)
  if (res[kResFailed] == 2.) return ctx->fail(LSA_E_GATE, "lsa_solve_device: the gate in front of the solve gave up waiting for the host");
  if (res[kResFailed] != 0.)
  {
    // iterations enqueued ahead are called off first (their gates wait for this thread, the stream would not drain)
    (void)lsa_icp_abandon(ctx);
    LSA_HIP(ctx, hipStreamSynchronize(ctx->stream));  // every block has given up or finished: the slots are quiet again
    ctx->lm_fallbacks++;
    return ctx->fail(LSA_E_STATE, "lsa_solve_device: a block waited too long for the others");
  }
  // the link this solve waited behind has served: its iteration ran, what its match announced stands
  if (waited_behind >= 0 && waited_behind < kGateRing && ctx->gate_saved[waited_behind].link) ctx->gate_saved[waited_behind].used = false;
  std::memset(out, 0, sizeof(*out));
  for (int a = 0; a < 6; ++a) out->pose[a] = res[kResPose + a];
  out->initial_cost = res[kResInitial];
  out->final_cost = res[kResFinal];
  out->cost = res[kResSums];
  for (int a = 0; a < 6; ++a) out->g[a] = res[kResSums + 1 + a];
  {
    int h = kResSums + 7;
    for (int a = 0; a < 6; ++a)
      for (int b = a; b < 6; ++b) { out->H[a * 6 + b] = res[h]; out->H[b * 6 + a] = res[h]; ++h; }
  }
  out->num_successful_steps = (int)res[kResSuccessful];
  out->num_unsuccessful_steps = (int)res[kResUnsuccessful];
  out->num_iterations = (int)res[kResIterations];
  out->num_evaluations = (int)res[kResEvaluations];
  out->skipped = (int)res[kResSkipped];
  out->termination = (int)res[kResCode];
  out->num_matches = (int)res[kResMatches];
  out->message = kMessages[std::min(std::max(out->termination, 0), 8)];
  // algorithmic bytes: every evaluation reads the record of every residual block (128 B + status)
  profile_add_bytes(ctx, "lm_solve", (double)out->num_evaluations * out->num_matches * 129);
  return LSA_OK;
}

int lsa_solve_device(lsa_ctx* ctx, unsigned type_mask, const double prior[6], int two_d_mode, int lm_max_iter, int min_matches, lsa_solve_result_t* out)
{
  if (!ctx || !prior || !out) return ctx ? ctx->fail(LSA_E_ARG, "lsa_solve_device: bad argument") : LSA_E_ARG;
  if (!ctx->lm_pending.empty()) return ctx->fail(LSA_E_STATE, "lsa_solve_device: another solve is in flight (lsa_solve_device_begin)");
  const int rc = lsa_solve_device_begin(ctx, type_mask, prior, two_d_mode, lm_max_iter, min_matches);
  if (rc) return rc;
  // host work the caller wants done while the kernel runs (one shot)
  if (ctx->solve_hook)
  {
    void (*fn)(void*) = ctx->solve_hook;
    ctx->solve_hook = nullptr;
    fn(ctx->solve_hook_arg);
  }
  return lsa_solve_device_end(ctx, out);
}

// ---- ICP iterations enqueued ahead of their inputs ------------------------------------------------------------------
// A gate is one small launch that waits, on the device, for the host to post what the launches behind it need (the pose
// the solve before it ended with: slam_lib/src/Slam.cxx:907, 940-946 and 1086, 1134-1142), or to call them off.  The
// kernel-launch path (a dozen microseconds a launch) is then not between the end of a solve and the next search: the
// launches are in the queue already and what remains is one store to coherent host memory and the gate's poll.
int lsa_icp_gate(lsa_ctx* ctx)
{
  if (!ctx) return LSA_E_ARG;
  if (!ctx->gate_host || !ctx->gate_dev) return ctx->fail(LSA_E_STATE, "lsa_icp_gate: no coherent host memory for the gates");
  LSA_HIP(ctx, hipSetDevice(ctx->device));
  const unsigned seq = ++ctx->gate_seq;
  const int ticket = (int)(seq % kGateRing);
  lsa_ctx::GateSaved& sv = ctx->gate_saved[ticket];
  sv = lsa_ctx::GateSaved();
  sv.used = true;
  sv.seq = seq;
  hipLaunchKernelGGL(k_icp_gate, dim3(1), dim3(kGateGranules), 0, ctx->stream, ctx->gate_host + (size_t)ticket * kGateGranules, ctx->gate_dev + (size_t)ticket * kGateWords, seq,
                     (ctx->debug_gate_give_up_every > 0 && seq % (unsigned)ctx->debug_gate_give_up_every == 0) ? 1 : 0);
  ctx->gate_current = ticket;
  return ticket;
}

// A link is a gate without the kernel and without the host: the block on the device is written by the SOLVE in front of it
// (k_lm_solve's leave_link: lsa_posemath.h's arithmetic, which the host repeats on the same result when it arrives), so that
// between two iterations there is one kernel boundary -- no trip over the bus, no host thread on the path.  The ticket is
// reserved here, handed to lsa_solve_device_begin_linked as the block to leave, and what is enqueued after that solve
// (lsa_match_types_gated, the next solve) waits behind it.
int lsa_icp_link(lsa_ctx* ctx)
{
  if (!ctx) return LSA_E_ARG;
  if (!ctx->gate_dev || !ctx->motion_dev) return ctx->fail(LSA_E_STATE, "lsa_icp_link: no device memory for the links");
  const unsigned seq = ctx->gate_seq + 1;
  const int ticket = (int)(seq % kGateRing);
  if (ctx->gate_saved[ticket].used) return ctx->fail(LSA_E_STATE, "lsa_icp_link: too many iterations enqueued ahead");
  ctx->gate_seq = seq;
  lsa_ctx::GateSaved& sv = ctx->gate_saved[ticket];
  sv = lsa_ctx::GateSaved();
  sv.used = true;
  sv.link = true;
  sv.seq = seq;
  return ticket;
}

// What the HOST works out between two iterations from a solve's result (lsa_posemath.h compiled for this side: the loops
// of host/lsa_slam_core.cpp call the same functions), laid out as the device's block: a test holds lsa_icp_link_peek
// against it, word for word.
int lsa_icp_link_expected(const double x[6], int skipped, int successful_steps, const lsa_icp_link_t* link, unsigned long long words[64], double motion_after[16])
{
  if (!x || !link || !words) return LSA_E_ARG;
  using namespace posemath;
  IcpGate g;
  std::memset(&g, 0, sizeof(g));
  std::memset(words, 0, kGateWords * sizeof(unsigned long long));
  if (motion_after) std::memcpy(motion_after, link->motion, 16 * sizeof(double));
  if (skipped || successful_steps == 1) return LSA_OK;  // go = 0: nothing else of the block means anything
  const Pose T = FromXYZRPY(x);
  ToXYZRPY(T, g.in.x0);
  g.in.pose = ToRigid(T);
  if (link->refine_undistortion)
  {
    WithinFrameMotion m;
    const double* mv = link->motion;
    m.Time0 = mv[0]; m.Time1 = mv[1];
    m.Rot0 = {mv[2], mv[3], mv[4], mv[5]};
    m.Rot1 = {mv[6], mv[7], mv[8], mv[9]};
    for (int i = 0; i < 3; ++i) { m.Trans0[i] = mv[10 + i]; m.Trans1[i] = mv[13 + i]; }
    ScanPoseClock clock;
    clock.have_log = link->have_log ? 1 : 0;
    clock.prev_time = link->prev_time;
    clock.cur_time = link->cur_time;
    clock.max_ratio = link->max_extrapolation_ratio;
    Pose previous, d0, d1;
    std::memcpy(previous.m, link->previous_world, sizeof(previous.m));
    // (the device keeps R and t of PreviousTworld: the last row is (0, 0, 0, 1) by construction)
    previous = FromRigid(ToRigid(previous));
    RefineUndistortion(m, clock, previous, T, d0, d1);
    g.in.ic = MakeInterpConst(d0, d1, m.Time0, m.Time1);
    if (motion_after)
    {
      double* o = motion_after;
      o[0] = m.Time0; o[1] = m.Time1;
      o[2] = m.Rot0.w; o[3] = m.Rot0.x; o[4] = m.Rot0.y; o[5] = m.Rot0.z;
      o[6] = m.Rot1.w; o[7] = m.Rot1.x; o[8] = m.Rot1.y; o[9] = m.Rot1.z;
      for (int i = 0; i < 3; ++i) { o[10 + i] = m.Trans0[i]; o[13 + i] = m.Trans1[i]; }
    }
  }
  g.go = 1ull;
  std::memcpy(words, &g, sizeof(g));
  return LSA_OK;
}

int lsa_icp_link_peek(lsa_ctx* ctx, int ticket, unsigned long long words[64])
{
  if (!ctx || !words || ticket < 0 || ticket >= kGateRing || !ctx->gate_dev) return ctx ? ctx->fail(LSA_E_ARG, "lsa_icp_link_peek: bad argument") : LSA_E_ARG;
  LSA_HIP(ctx, hipSetDevice(ctx->device));
  LSA_HIP(ctx, hipStreamSynchronize(ctx->stream));
  LSA_HIP(ctx, hipMemcpy(words, ctx->gate_dev + (size_t)ticket * kGateWords, kGateWords * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  return LSA_OK;
}

static int gate_release(lsa_ctx* ctx, int ticket, const IcpGate* block)
{
  if (!ctx || ticket < 0 || ticket >= kGateRing || !ctx->gate_host || !ctx->gate_saved[ticket].used)
    return ctx ? ctx->fail(LSA_E_ARG, "lsa_icp_post / lsa_icp_cancel: no such gate") : LSA_E_ARG;
  unsigned long long* g = ctx->gate_host + (size_t)ticket * kGateGranules;
  // which gate this slot serves now: the newest one enqueued with this ticket
  unsigned seq = ctx->gate_seq;
  while ((int)(seq % kGateRing) != ticket) --seq;
  unsigned long long words[kGateWords] = {};
  if (block) std::memcpy(words, block, sizeof(IcpGate));  // (word 0 = go = 1; called off: all zero)
  // every granule carries the gate's number: the device has the block once it has seen it in all of them
  for (int i = 0; i < kGateWords; ++i)
  {
    __atomic_store_n(g + 2 * i, ((unsigned long long)seq << 32) | (words[i] & 0xffffffffull), __ATOMIC_RELAXED);
    __atomic_store_n(g + 2 * i + 1, ((unsigned long long)seq << 32) | (words[i] >> 32), __ATOMIC_RELAXED);
  }
  if (lsa_icp_trace_on()) std::fprintf(stderr, "[gate] ticket %d seq %u %s\n", ticket, seq, block ? "posted" : "called off");
  ctx->gate_saved[ticket].used = false;
  if (ctx->gate_current == ticket) ctx->gate_current = -1;
  return LSA_OK;
}

int lsa_icp_post(lsa_ctx* ctx, int ticket, const double pose[16], const double prior[6], const double H0[16], const double H1[16], double t0, double t1)
{
  if (!ctx || !pose || !prior || ((H0 == nullptr) != (H1 == nullptr))) return ctx ? ctx->fail(LSA_E_ARG, "lsa_icp_post: bad argument") : LSA_E_ARG;
  IcpGate g;
  std::memset(&g, 0, sizeof(g));
  g.go = 1;
  row_major_to_rt(pose, g.in.pose.R, g.in.pose.t);
  for (int a = 0; a < 6; ++a) g.in.x0[a] = prior[a];
  if (H0) g.in.ic = make_interp_const(H0, H1, t0, t1);
  return gate_release(ctx, ticket, &g);
}

int lsa_icp_cancel(lsa_ctx* ctx, int ticket)
{
  if (!ctx || ticket < 0 || ticket >= kGateRing) return LSA_E_ARG;
  // what the launches behind the gate had announced on the host is taken back: they will not run
  const lsa_ctx::GateSaved sv = ctx->gate_saved[ticket];
  if (sv.link)
  {
    // (the device has called it off itself -- the solve in front left go = 0 -- or will never reach it)
    if (!sv.used) return ctx->fail(LSA_E_ARG, "lsa_icp_cancel: no such link");
    ctx->gate_saved[ticket].used = false;
    if (ctx->gate_current == ticket) ctx->gate_current = -1;
  }
  else
  {
    const int rc = gate_release(ctx, ticket, nullptr);
    if (rc) return rc;
  }
  for (int k = 0; k < 3; ++k)
    if ((sv.mask >> k) & 1u)
    {
      ctx->match[k].sat = sv.sat[k];
      ctx->match[k].k = sv.k[k];
      ctx->match[k].valid = sv.valid[k];
      ctx->hist_pos[k] = sv.hist_pos[k];
      ctx->hist_serial[k] = sv.hist_serial[k];
    }
  return LSA_OK;
}

int lsa_icp_abandon(lsa_ctx* ctx)
{
  if (!ctx) return LSA_E_ARG;
  // the youngest first: each takes back what ITS match announced, the oldest one's "before" is what remains
  while (true)
  {
    int youngest = -1;
    for (int t = 0; t < kGateRing; ++t)
      if (ctx->gate_saved[t].used && (youngest < 0 || (int)(ctx->gate_saved[t].seq - ctx->gate_saved[youngest].seq) > 0)) youngest = t;
    if (youngest < 0) break;
    if (lsa_icp_cancel(ctx, youngest) != LSA_OK) ctx->gate_saved[youngest].used = false;
  }
  ctx->lm_pending.clear();
  ctx->lm_pending_wait.clear();
  ctx->gate_current = -1;
  return LSA_OK;
}

int lsa_debug_set(lsa_ctx* ctx, const char* name, int value)
{
  if (!ctx || !name) return LSA_E_ARG;
  const std::string n(name);
  if (n == "gate_give_up_every") ctx->debug_gate_give_up_every = value;
  else if (n == "lm_give_up_block") ctx->debug_lm_give_up_block = value;
  else return ctx->fail(LSA_E_ARG, "lsa_debug_set: no such knob");
  return LSA_OK;
}

int lsa_solve_device_fallbacks(const lsa_ctx* ctx) { return ctx ? ctx->lm_fallbacks : 0; }

int lsa_solve_device_interlude(lsa_ctx* ctx, void (*fn)(void*), void* arg)
{
  if (!ctx) return LSA_E_ARG;
  ctx->solve_hook = fn;
  ctx->solve_hook_arg = arg;
  return LSA_OK;
}

int lsa_solve_device_trace(lsa_ctx* ctx, unsigned long long out[12])
{
  if (!ctx || !out || !ctx->trace_dev) return LSA_E_ARG;
  if (hipStreamSynchronize(ctx->stream) != hipSuccess) return LSA_E_HIP;
  if (hipMemcpy(out, reinterpret_cast<unsigned long long*>(ctx->trace_dev) + (size_t)8192 * 12, 12 * sizeof(unsigned long long), hipMemcpyDeviceToHost) != hipSuccess) return LSA_E_HIP;
  return LSA_OK;
}

}  // extern "C"
