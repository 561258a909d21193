// lsa_accum.h -- what Ceres evaluates per LM step for one residual block, shared by the two kernels
// that evaluate it: k_accumulate (one evaluation per launch, host-driven trust region, lsa_match.hip)
// and k_lm_solve (the whole LocalOptimizer::Solve in one launch, lsa_lm.hip).
//
//   residual   r = A (R(rpy) X + t - P)              slam_lib/include/LidarSlam/CeresCostFunctions.h:105-152
//   rotation   R = Rz(rz) Ry(ry) Rx(rx)              CeresCostFunctions.h:67-79
//   loss       ScaledLoss(TukeyLoss(a), weight)      slam_lib/src/KeypointsMatcher.cxx:84-101 (Ceres >= 2)
//   Jacobian   analytic: [A, A dR/drx X, A dR/dry X, A dR/drz X]  (replaces the Jet<double, 6> autodiff)
// Since rho'' <= 0 the Triggs corrector only scales r and J by sqrt(rho'): H = sum rho' J^T J,
// g = sum rho' J^T r, cost = 1/2 sum rho.
#pragma once
#include "lsa_ctx.h"

namespace lsa
{

// the point of the evaluation: only ever indexed with constants (registers)
struct RotConst
{
  double R[9], dRx[9], dRy[9], dRz[9];
  double t[3];
};
// the residual blocks, per keypoint type: indexed with the type at run time, so it has to live in the kernel
// arguments (a copy in registers would be demoted to scratch memory)
struct RecordSet
{
  const double* rec[3];
  const uint8_t* status[3];
  int count[3];
  int cap[3];
  double sat2[3];  // Tukey a^2 per type
};
struct AccumConst
{
  RotConst rot;
  RecordSet set;
  int jac;
};

// R = Rz Ry Rx and its partial derivatives from the six parameters (x, y, z, rx, ry, rz) given the sines and
// cosines of the three angles
__host__ __device__ inline void rotation_and_derivatives(double cx, double sx, double cy, double sy, double cz, double sz, double R[9], double dRx[9],
                                                         double dRy[9], double dRz[9])
{
  R[0] = cy * cz; R[1] = sx * sy * cz - cx * sz; R[2] = cx * sy * cz + sx * sz;
  R[3] = cy * sz; R[4] = sx * sy * sz + cx * cz; R[5] = cx * sy * sz - sx * cz;
  R[6] = -sy; R[7] = sx * cy; R[8] = cx * cy;
  dRx[0] = 0; dRx[1] = cx * sy * cz + sx * sz; dRx[2] = -sx * sy * cz + cx * sz;
  dRx[3] = 0; dRx[4] = cx * sy * sz - sx * cz; dRx[5] = -sx * sy * sz - cx * cz;
  dRx[6] = 0; dRx[7] = cx * cy; dRx[8] = -sx * cy;
  dRy[0] = -sy * cz; dRy[1] = sx * cy * cz; dRy[2] = cx * cy * cz;
  dRy[3] = -sy * sz; dRy[4] = sx * cy * sz; dRy[5] = cx * cy * sz;
  dRy[6] = -cy; dRy[7] = -sx * sy; dRy[8] = -cx * sy;
  dRz[0] = -cy * sz; dRz[1] = -sx * sy * sz - cx * cz; dRz[2] = -cx * sy * sz + sx * cz;
  dRz[3] = cy * cz; dRz[4] = sx * sy * cz - cx * sz; dRz[5] = cx * sy * cz + sx * sz;
  dRz[6] = 0; dRz[7] = 0; dRz[8] = 0;
}

#if defined(__HIPCC__)
__device__ __forceinline__ void mv3(const double M[9], double x, double y, double z, double& ox, double& oy, double& oz)
{
  ox = (M[0] * x + M[1] * y) + M[2] * z;
  oy = (M[3] * x + M[4] * y) + M[5] * z;
  oz = (M[6] * x + M[7] * y) + M[8] * z;
}

// x of this lane's half (rows) + x of the partner half (rows): first operand + second operand, as lane i < W gets
// from a shuffle-down by W
template <int W>
__device__ __forceinline__ double swap_add(double x)
{
  const long long b = __double_as_longlong(x);
  const unsigned xl = (unsigned)(b & 0xffffffffll), xh = (unsigned)((unsigned long long)b >> 32);
  const auto lo = W == 32 ? __builtin_amdgcn_permlane32_swap(xl, xl, false, false) : __builtin_amdgcn_permlane16_swap(xl, xl, false, false);
  const auto hi = W == 32 ? __builtin_amdgcn_permlane32_swap(xh, xh, false, false) : __builtin_amdgcn_permlane16_swap(xh, xh, false, false);
  const double a = __longlong_as_double((long long)(((unsigned long long)hi[0] << 32) | lo[0]));  // lower half / even rows, everywhere
  const double c = __longlong_as_double((long long)(((unsigned long long)hi[1] << 32) | lo[1]));  // upper half / odd rows, everywhere
  return a + c;
}

// value of lane (i + o) of the same row of 16 lanes, 0 where that lane does not exist (o = 1, 2, 4, 8)
__device__ __forceinline__ double dpp_row_shl(double x, int o)
{
  const long long b = __double_as_longlong(x);
  int lo = (int)(b & 0xffffffffll), hi = (int)(b >> 32);
  switch (o)
  {
    case 8: lo = __builtin_amdgcn_update_dpp(0, lo, 0x108, 0xF, 0xF, true); hi = __builtin_amdgcn_update_dpp(0, hi, 0x108, 0xF, 0xF, true); break;
    case 4: lo = __builtin_amdgcn_update_dpp(0, lo, 0x104, 0xF, 0xF, true); hi = __builtin_amdgcn_update_dpp(0, hi, 0x104, 0xF, 0xF, true); break;
    case 2: lo = __builtin_amdgcn_update_dpp(0, lo, 0x102, 0xF, 0xF, true); hi = __builtin_amdgcn_update_dpp(0, hi, 0x102, 0xF, 0xF, true); break;
    default: lo = __builtin_amdgcn_update_dpp(0, lo, 0x101, 0xF, 0xF, true); hi = __builtin_amdgcn_update_dpp(0, hi, 0x101, 0xF, 0xF, true); break;
  }
  return __longlong_as_double(((long long)hi << 32) | (long long)(unsigned)lo);
}

// one residual block's contribution: acc += its cost, gradient, J^T J and count
__device__ __forceinline__ void accumulate_one(const double A[9], double Px, double Py, double Pz, double Xx, double Xy, double Xz, double weight, double a2,
                                               const RotConst& c, bool jac, double acc[kAccumVals])
{
  double yx, yy, yz;
  mv3(c.R, Xx, Xy, Xz, yx, yy, yz);
  const double dx = (yx + c.t[0]) - Px, dy = (yy + c.t[1]) - Py, dz = (yz + c.t[2]) - Pz;
  double r0, r1, r2;
  mv3(A, dx, dy, dz, r0, r1, r2);
  const double s = (r0 * r0 + r1 * r1) + r2 * r2;
  double rho0, rho1;
  if (s <= a2)
  {
    const double value = 1.0 - s / a2;
    const double value_sq = value * value;
    rho0 = a2 / 3.0 * (1.0 - value_sq * value);
    rho1 = value_sq;
  }
  else { rho0 = a2 / 3.0; rho1 = 0.0; }
  rho0 *= weight; rho1 *= weight;
  acc[0] += 0.5 * rho0;
  acc[28] += 1.0;
  if (!jac) return;
  double J[3][6];
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int b = 0; b < 3; ++b) J[a][b] = A[a * 3 + b];
  double vx, vy, vz, cx, cy, cz;
  mv3(c.dRx, Xx, Xy, Xz, vx, vy, vz); mv3(A, vx, vy, vz, cx, cy, cz); J[0][3] = cx; J[1][3] = cy; J[2][3] = cz;
  mv3(c.dRy, Xx, Xy, Xz, vx, vy, vz); mv3(A, vx, vy, vz, cx, cy, cz); J[0][4] = cx; J[1][4] = cy; J[2][4] = cz;
  mv3(c.dRz, Xx, Xy, Xz, vx, vy, vz); mv3(A, vx, vy, vz, cx, cy, cz); J[0][5] = cx; J[1][5] = cy; J[2][5] = cz;
  int h = 7;
#pragma unroll
  for (int a = 0; a < 6; ++a)
  {
    acc[1 + a] += rho1 * ((J[0][a] * r0 + J[1][a] * r1) + J[2][a] * r2);
#pragma unroll
    for (int b = a; b < 6; ++b) acc[h++] += rho1 * ((J[0][a] * J[0][b] + J[1][a] * J[1][b]) + J[2][a] * J[2][b]);
  }
}

// this thread's share of the residual blocks (grid stride over EDGE[0..), PLANE[0..), BLOB[0..)): acc += their
// cost, gradient, J^T J and count
__device__ __forceinline__ void accumulate_records(const RecordSet& rs, const RotConst& c, bool jac, int first, int stride, double acc[kAccumVals])
{
  const int total = rs.count[0] + rs.count[1] + rs.count[2];
  for (int gidx = first; gidx < total; gidx += stride)
  {
    int t = 0, i = gidx;
    if (i >= rs.count[0]) { i -= rs.count[0]; t = 1; if (i >= rs.count[1]) { i -= rs.count[1]; t = 2; } }
    if (rs.status[t][i] != LSA_MATCH_SUCCESS) continue;
    const double* rec = rs.rec[t];
    const size_t cap = (size_t)rs.cap[t];
    double A[9];
#pragma unroll
    for (int f = 0; f < 9; ++f) A[f] = rec[f * cap + i];
    const double Px = rec[9 * cap + i], Py = rec[10 * cap + i], Pz = rec[11 * cap + i];
    const double Xx = rec[12 * cap + i], Xy = rec[13 * cap + i], Xz = rec[14 * cap + i];
    const double weight = rec[15 * cap + i];
    accumulate_one(A, Px, Py, Pz, Xx, Xy, Xz, weight, rs.sat2[t], c, jac, acc);
  }
}

// A thread's FIRST residual block on its way from memory into registers (record_load: the loads are issued, nothing waits
// for them) and from there into the cache of accumulate_records_cached (record_stash).  The solve kernel issues the loads as
// its very first instructions: what it has to wait for before the first evaluation anyway -- the word that says whether the
// launch runs at all, the start point, its sines and cosines -- then costs no time of its own.
struct RecordRegs
{
  double v[16];
  double a2;
  int state;  // 1 a residual block, 0 a rejected keypoint, -1 nothing (no residual block for this thread)
};
__device__ __forceinline__ void record_load(const RecordSet& rs, int gidx, RecordRegs& r)
{
  const int total = rs.count[0] + rs.count[1] + rs.count[2];
  r.state = -1;
  r.a2 = 0.;
#pragma unroll
  for (int f = 0; f < 16; ++f) r.v[f] = 0.;
  if (gidx >= total) return;
  int t = 0, i = gidx;
  if (i >= rs.count[0]) { i -= rs.count[0]; t = 1; if (i >= rs.count[1]) { i -= rs.count[1]; t = 2; } }
  const double* rec = rs.rec[t];
  const size_t cap = (size_t)rs.cap[t];
  const uint8_t st = rs.status[t][i];
  // (the record of a rejected keypoint is memory like any other: read, not used)
#pragma unroll
  for (int f = 0; f < 16; ++f) r.v[f] = rec[f * cap + i];
  r.a2 = rs.sat2[t];
  r.state = st == LSA_MATCH_SUCCESS ? 1 : 0;
}
__device__ __forceinline__ void record_stash(const RecordRegs& r, double* __restrict__ cache, int cstride, int at)
{
  if (r.state == 1)
  {
#pragma unroll
    for (int f = 0; f < 16; ++f) cache[f * cstride + at] = r.v[f];
    cache[16 * cstride + at] = r.a2;
  }
  else if (r.state == 0)
    cache[15 * cstride + at] = -1.;
}

// The same with the thread's first `cslots` residual blocks kept in LDS between the evaluations of one launch
// (cache[f * cstride + slot * blockDim.x + thread], f = 0..16: A, P, X, weight, a^2; weight < 0 marks a rejected keypoint):
// the first evaluation (fill == true) reads global memory and fills the cache -- but for the first `prefilled` slots, which are
// there already (record_stash) --, the others read the cache.  Same records, same order, same arithmetic as accumulate_records.
__device__ __forceinline__ void accumulate_records_cached(const RecordSet& rs, const RotConst& c, int first, int stride, double* __restrict__ cache,
                                                          int cslots, int cstride, bool fill, int prefilled, double acc[kAccumVals])
{
  const int total = rs.count[0] + rs.count[1] + rs.count[2];
  int j = 0;
  for (int gidx = first; gidx < total; gidx += stride, ++j)
  {
    double A[9], Px, Py, Pz, Xx, Xy, Xz, weight, a2;
    const int at = j * (int)blockDim.x + (int)threadIdx.x;
    if (j < cslots && (!fill || j < prefilled))
    {
      weight = cache[15 * cstride + at];
      if (weight < 0.) continue;
#pragma unroll
      for (int f = 0; f < 9; ++f) A[f] = cache[f * cstride + at];
      Px = cache[9 * cstride + at]; Py = cache[10 * cstride + at]; Pz = cache[11 * cstride + at];
      Xx = cache[12 * cstride + at]; Xy = cache[13 * cstride + at]; Xz = cache[14 * cstride + at];
      a2 = cache[16 * cstride + at];
    }
    else
    {
      int t = 0, i = gidx;
      if (i >= rs.count[0]) { i -= rs.count[0]; t = 1; if (i >= rs.count[1]) { i -= rs.count[1]; t = 2; } }
      if (rs.status[t][i] != LSA_MATCH_SUCCESS)
      {
        if (j < cslots) cache[15 * cstride + at] = -1.;
        continue;
      }
      const double* rec = rs.rec[t];
      const size_t cap = (size_t)rs.cap[t];
#pragma unroll
      for (int f = 0; f < 9; ++f) A[f] = rec[f * cap + i];
      Px = rec[9 * cap + i]; Py = rec[10 * cap + i]; Pz = rec[11 * cap + i];
      Xx = rec[12 * cap + i]; Xy = rec[13 * cap + i]; Xz = rec[14 * cap + i];
      weight = rec[15 * cap + i];
      a2 = rs.sat2[t];
      if (j < cslots)
      {
#pragma unroll
        for (int f = 0; f < 9; ++f) cache[f * cstride + at] = A[f];
        cache[9 * cstride + at] = Px; cache[10 * cstride + at] = Py; cache[11 * cstride + at] = Pz;
        cache[12 * cstride + at] = Xx; cache[13 * cstride + at] = Xy; cache[14 * cstride + at] = Xz;
        cache[15 * cstride + at] = weight; cache[16 * cstride + at] = a2;
      }
    }
    accumulate_one(A, Px, Py, Pz, Xx, Xy, Xz, weight, a2, c, true, acc);
  }
}

// Fixed-order reduction of the 29 sums over the wavefront, TRANSPOSED: every step halves the number of values a lane
// carries instead of carrying all of them through all six steps -- 32 exchanges and additions in place of 174.
//   halves of the wavefront: v_permlane32_swap on the pair (value i, value 16 + i) hands the lower half both halves'
//     value i and the upper half both halves' value 16 + i in one VALU pass per word: 16 values left per lane;
//   rows of 16 lanes: v_permlane16_swap likewise on (i, 8 + i): 8 values left, row r holding values 8 r .. 8 r + 7;
//   inside a row: partners at lane distance 8 (row_ror:8), then the mirror image inside 8 lanes (row_half_mirror),
//     then distance 2 and 1 (quad_perm): 4, 2, 1 values left, a lane keeps the half its own bit selects and gets the
//     partner's part of that half through a DPP operand.
// Afterwards lane L holds the wavefront's total of value 8 (L >> 4) + ((L >> 1) & 7) (odd lanes duplicate their even
// neighbours); the summation order is fixed by the lane number alone.
template <int CTRL>
__device__ __forceinline__ double dpp_mov(double x)
{
  const long long b = __double_as_longlong(x);
  const int lo = __builtin_amdgcn_update_dpp(0, (int)(b & 0xffffffffll), CTRL, 0xF, 0xF, true);
  const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), CTRL, 0xF, 0xF, true);
  return __longlong_as_double(((long long)hi << 32) | (long long)(unsigned)lo);
}
// x = value for the lower half / even rows, y = value for the upper half / odd rows: the sum over both halves (rows) of
// the one this lane's half (row) keeps
template <int W>
__device__ __forceinline__ double swap_keep_add(double x, double y)
{
  const long long bx = __double_as_longlong(x), by = __double_as_longlong(y);
  const unsigned xl = (unsigned)(bx & 0xffffffffll), xh = (unsigned)((unsigned long long)bx >> 32);
  const unsigned yl = (unsigned)(by & 0xffffffffll), yh = (unsigned)((unsigned long long)by >> 32);
  // swap(a, b): a's upper half (odd rows) <-> b's lower half (even rows); afterwards [0] = {a.lower, b.lower}, [1] = {a.upper, b.upper}
  const auto lo = W == 32 ? __builtin_amdgcn_permlane32_swap(xl, yl, false, false) : __builtin_amdgcn_permlane16_swap(xl, yl, false, false);
  const auto hi = W == 32 ? __builtin_amdgcn_permlane32_swap(xh, yh, false, false) : __builtin_amdgcn_permlane16_swap(xh, yh, false, false);
  const double a = __longlong_as_double((long long)(((unsigned long long)hi[0] << 32) | lo[0]));
  const double c = __longlong_as_double((long long)(((unsigned long long)hi[1] << 32) | lo[1]));
  return a + c;
}
__device__ __forceinline__ double wave_reduce_accum(const double acc[kAccumVals], int& slot)
{
  static_assert(kAccumVals <= 32, "32 slots");
  const int lane = threadIdx.x & 63;
  double a[16], b[8], c[4], d[2];
#pragma unroll
  for (int i = 0; i < 16; ++i) a[i] = swap_keep_add<32>(acc[i], 16 + i < kAccumVals ? acc[16 + i] : 0.);
#pragma unroll
  for (int i = 0; i < 8; ++i) b[i] = swap_keep_add<16>(a[i], a[8 + i]);
  const bool b3 = (lane & 8) != 0, b2 = (lane & 4) != 0, b1 = (lane & 2) != 0;
#pragma unroll
  for (int i = 0; i < 4; ++i)
  {
    const double keep = b3 ? b[4 + i] : b[i], send = b3 ? b[i] : b[4 + i];
    c[i] = keep + dpp_mov<0x128>(send);  // row_ror:8
  }
#pragma unroll
  for (int i = 0; i < 2; ++i)
  {
    const double keep = b2 ? c[2 + i] : c[i], send = b2 ? c[i] : c[2 + i];
    d[i] = keep + dpp_mov<0x141>(send);  // row_half_mirror: lane j <-> 7 - j of its 8, the other value of bit 2
  }
  double e;
  {
    const double keep = b1 ? d[1] : d[0], send = b1 ? d[0] : d[1];
    e = keep + dpp_mov<0x4E>(send);      // quad_perm [2, 3, 0, 1]
  }
  e = e + dpp_mov<0xB1>(e);              // quad_perm [1, 0, 3, 2]
  // which value this lane ended up with: the half-mirror step pairs j with 7 - j, so from there on the lane's bits 1 and 0
  // are those of min(j, 7 - j) in its group of four ... the value index only depends on the bits that SELECTED
  slot = 8 * (lane >> 4) + (b3 ? 4 : 0) + (b2 ? 2 : 0) + (b1 ? 1 : 0);
  return e;
}
#endif  // __HIPCC__

}  // namespace lsa
