// lsa_transform.hip -- undistortion and rigid / interpolated transforms of device point sets.
//   k_undistort          Slam::RefineUndistortion per-point loop (slam_lib/src/Slam.cxx:1342-1351)
//   k_transform_out      Slam::TransformPointCloud (:1491-1509) / AggregateFrames(..., true) (:1512-1578)
//   k_time_range, k_bbox Slam::InitUndistortion (:1291-1300), getMinMax3D of the world keypoints (:1026-1029)
// All arithmetic in double, stored as float (Utils::TransformPoint, Utilities.h:274-278).
#include <cfloat>
#include <cmath>
#include <limits>
#include "lsa_ctx.h"
#include "lsa_device_math.h"

using namespace lsa;

namespace
{

__device__ __forceinline__ double point_time(const float4& b) { return __hiloint2double(__float_as_int(b.y), __float_as_int(b.x)); }

// in place, the three keypoint types in one launch (blockIdx.y = type)
struct KpSetsRW
{
  float4* pts[3];
  int n[3];
};
__global__ __launch_bounds__(256) void k_undistort(KpSetsRW s, InterpConst c)
{
  float4* __restrict__ pts = s.pts[blockIdx.y];
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= s.n[blockIdx.y]) return;
  float4 a = pts[2 * (size_t)i];
  const float4 b = pts[2 * (size_t)i + 1];
  Rigid T;
  interp_eval(c, point_time(b), T);
  double ox, oy, oz;
  rigid_apply(T, (double)a.x, (double)a.y, (double)a.z, ox, oy, oz);
  a.x = (float)ox; a.y = (float)oy; a.z = (float)oz;
  pts[2 * (size_t)i] = a;
}

// out = T(point) for a whole set; interpolate != 0 -> per-point pose
__global__ __launch_bounds__(256) void k_transform_out(const float4* __restrict__ in, int n, int interpolate, InterpConst c, Rigid R,
                                                       double time_offset, float4* __restrict__ out)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float4 a = in[2 * (size_t)i];
  float4 b = in[2 * (size_t)i + 1];
  if (time_offset != 0.)
  {
    // point.time += timeOffset, before the pose is evaluated at it (Slam.cxx:1547-1549)
    const double t = point_time(b) + time_offset;
    b.x = __int_as_float(__double2loint(t));
    b.y = __int_as_float(__double2hiint(t));
  }
  Rigid T = R;
  if (interpolate) interp_eval(c, point_time(b), T);
  double ox, oy, oz;
  rigid_apply(T, (double)a.x, (double)a.y, (double)a.z, ox, oy, oz);
  a.x = (float)ox; a.y = (float)oy; a.z = (float)oz;
  out[2 * (size_t)i] = a;
  out[2 * (size_t)i + 1] = b;
}

// the three keypoint types of a set in one launch, written straight into pinned host memory (blockIdx.y = type)
struct StageOut
{
  const float4* in[3];
  float4* out[3];
  int n[3];
};
__global__ __launch_bounds__(256) void k_transform_stage(StageOut s, Rigid T)
{
  const int t = blockIdx.y;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= s.n[t]) return;
  float4 a = s.in[t][2 * (size_t)i];
  const float4 b = s.in[t][2 * (size_t)i + 1];
  double ox, oy, oz;
  rigid_apply(T, (double)a.x, (double)a.y, (double)a.z, ox, oy, oz);
  a.x = (float)ox; a.y = (float)oy; a.z = (float)oz;
  s.out[t][2 * (size_t)i] = a;
  s.out[t][2 * (size_t)i + 1] = b;
}

__global__ __launch_bounds__(256) void k_copy_sets(StageOut s)
{
  const int t = blockIdx.y;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= s.n[t]) return;
  s.out[t][2 * (size_t)i] = s.in[t][2 * (size_t)i];
  s.out[t][2 * (size_t)i + 1] = s.in[t][2 * (size_t)i + 1];
}

// monotone encoding of doubles for 64-bit atomicMin/Max
__device__ __forceinline__ unsigned long long d2o(double d)
{
  unsigned long long u = (unsigned long long)__double_as_longlong(d);
  return (u & 0x8000000000000000ull) ? ~u : (u | 0x8000000000000000ull);
}
__host__ inline double o2d_host(unsigned long long u)
{
  u = (u & 0x8000000000000000ull) ? (u & 0x7fffffffffffffffull) : ~u;
  double d;
  std::memcpy(&d, &u, sizeof(d));
  return d;
}

__global__ __launch_bounds__(256) void k_time_range(const float4* __restrict__ pts, int n, unsigned long long* __restrict__ bits)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  unsigned long long lo = ~0ull, hi = 0ull;
  if (i < n)
  {
    const unsigned long long o = d2o(point_time(pts[2 * (size_t)i + 1]));
    lo = hi = o;
  }
  for (int s = 32; s > 0; s >>= 1)
  {
    const unsigned long long l2 = __shfl_down(lo, s), h2 = __shfl_down(hi, s);
    lo = l2 < lo ? l2 : lo;
    hi = h2 > hi ? h2 : hi;
  }
  if ((threadIdx.x & 63) == 0 && i < n + 64)
  {
    atomicMin(&bits[0], lo);
    atomicMax(&bits[1], hi);
  }
}

__device__ __forceinline__ unsigned f2ou(float f)
{
  unsigned u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__host__ inline float ou2f_host(unsigned u)
{
  u = (u & 0x80000000u) ? (u & 0x7fffffffu) : ~u;
  float f;
  std::memcpy(&f, &u, sizeof(f));
  return f;
}

__global__ __launch_bounds__(256) void k_bbox(const float4* __restrict__ pts, int n, Rigid T, unsigned* __restrict__ bits)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  unsigned lo[3] = {~0u, ~0u, ~0u}, hi[3] = {0u, 0u, 0u};
  if (i < n)
  {
    const float4 a = pts[2 * (size_t)i];
    double ox, oy, oz;
    rigid_apply(T, (double)a.x, (double)a.y, (double)a.z, ox, oy, oz);
    const float v[3] = {(float)ox, (float)oy, (float)oz};
    for (int d = 0; d < 3; ++d) lo[d] = hi[d] = f2ou(v[d]);
  }
  for (int d = 0; d < 3; ++d)
    for (int s = 32; s > 0; s >>= 1)
    {
      const unsigned l2 = __shfl_down(lo[d], s), h2 = __shfl_down(hi[d], s);
      lo[d] = l2 < lo[d] ? l2 : lo[d];
      hi[d] = h2 > hi[d] ? h2 : hi[d];
    }
  if ((threadIdx.x & 63) == 0)
    for (int d = 0; d < 3; ++d)
    {
      atomicMin(&bits[d], lo[d]);
      atomicMax(&bits[3 + d], hi[d]);
    }
}

// the three keypoint sets at once: blockIdx.y = type, counts read from the device (they may not be on the
// host yet), one set of atomics per block.  bits: [0..1] time range, [2 + 6 * type ..] bounding boxes.
struct KpSets
{
  const float4* pts[3];
  int n[3];              // used when counts == nullptr
};
__global__ void k_range_init(unsigned long long* __restrict__ bits)
{
  if (threadIdx.x == 0) { bits[0] = ~0ull; bits[1] = 0ull; }
}
__global__ void k_bbox_init(unsigned* __restrict__ b32)
{
  if (threadIdx.x < 18) b32[threadIdx.x] = (threadIdx.x % 6) < 3 ? ~0u : 0u;
}
__global__ __launch_bounds__(256) void k_time_range3(KpSets sets, const int* __restrict__ counts, unsigned long long* __restrict__ bits)
{
  __shared__ unsigned long long slo[4], shi[4];
  const int t = blockIdx.y;
  const int n = counts ? counts[t] : sets.n[t];
  if ((int)(blockIdx.x * blockDim.x) >= n) return;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  unsigned long long lo = ~0ull, hi = 0ull;
  if (i < n) lo = hi = d2o(point_time(sets.pts[t][2 * (size_t)i + 1]));
  for (int s = 32; s > 0; s >>= 1)
  {
    const unsigned long long l2 = __shfl_down(lo, s), h2 = __shfl_down(hi, s);
    lo = l2 < lo ? l2 : lo;
    hi = h2 > hi ? h2 : hi;
  }
  if ((threadIdx.x & 63) == 0) { slo[threadIdx.x >> 6] = lo; shi[threadIdx.x >> 6] = hi; }
  __syncthreads();
  if (threadIdx.x == 0)
  {
    for (int w = 1; w < 4; ++w) { lo = slo[w] < lo ? slo[w] : lo; hi = shi[w] > hi ? shi[w] : hi; }
    atomicMin(&bits[0], lo);
    atomicMax(&bits[1], hi);
  }
}
__global__ __launch_bounds__(256) void k_bbox3(KpSets sets, Rigid T0, int interpolate, InterpConst c, unsigned* __restrict__ bits)
{
  __shared__ unsigned slo[4][3], shi[4][3];
  const int t = blockIdx.y;
  const int n = sets.n[t];
  if ((int)(blockIdx.x * blockDim.x) >= n) return;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  unsigned lo[3] = {~0u, ~0u, ~0u}, hi[3] = {0u, 0u, 0u};
  if (i < n)
  {
    const float4 a = sets.pts[t][2 * (size_t)i];
    Rigid T = T0;
    if (interpolate) interp_eval(c, point_time(sets.pts[t][2 * (size_t)i + 1]), T);  // the pose at the point's own time
    double ox, oy, oz;
    rigid_apply(T, (double)a.x, (double)a.y, (double)a.z, ox, oy, oz);
    const float v[3] = {(float)ox, (float)oy, (float)oz};
    // a NaN coordinate takes no part, as in a min / max loop written with comparisons
    for (int d = 0; d < 3; ++d)
      if (v[d] == v[d]) lo[d] = hi[d] = f2ou(v[d]);
  }
  for (int d = 0; d < 3; ++d)
    for (int s = 32; s > 0; s >>= 1)
    {
      const unsigned l2 = __shfl_down(lo[d], s), h2 = __shfl_down(hi[d], s);
      lo[d] = l2 < lo[d] ? l2 : lo[d];
      hi[d] = h2 > hi[d] ? h2 : hi[d];
    }
  if ((threadIdx.x & 63) == 0)
    for (int d = 0; d < 3; ++d) { slo[threadIdx.x >> 6][d] = lo[d]; shi[threadIdx.x >> 6][d] = hi[d]; }
  __syncthreads();
  if (threadIdx.x < 3)
  {
    const int d = threadIdx.x;
    unsigned l = slo[0][d], h = shi[0][d];
    for (int w = 1; w < 4; ++w) { l = slo[w][d] < l ? slo[w][d] : l; h = shi[w][d] > h ? shi[w][d] : h; }
    atomicMin(&bits[6 * t + d], l);
    atomicMax(&bits[6 * t + 3 + d], h);
  }
}

// What the localization starts with, in ONE launch: working = the raw keypoints, undistorted (the reset and the first
// RefineUndistortion of Slam::Localization, Slam.cxx:980-999), and the bounding boxes of the three types under the pose
// guess (pcl::getMinMax3D of the transformed keypoints, Slam.cxx:1027-1029).  The boxes' words must be armed (k_bbox_init)
// before the launch: the blocks reduce into them with atomics.
__global__ __launch_bounds__(256) void k_loc_start(StageOut s, int undistort, InterpConst c, int boxes, Rigid T0, unsigned* __restrict__ bits)
{
  __shared__ unsigned slo[4][3], shi[4][3];
  const int t = blockIdx.y;
  const int n = s.n[t];
  if ((int)(blockIdx.x * blockDim.x) >= n) return;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  unsigned lo[3] = {~0u, ~0u, ~0u}, hi[3] = {0u, 0u, 0u};
  if (i < n)
  {
    float4 a = s.in[t][2 * (size_t)i];
    const float4 b = s.in[t][2 * (size_t)i + 1];
    if (undistort)
    {
      Rigid U;
      interp_eval(c, point_time(b), U);
      double ox, oy, oz;
      rigid_apply(U, (double)a.x, (double)a.y, (double)a.z, ox, oy, oz);
      a.x = (float)ox; a.y = (float)oy; a.z = (float)oz;
    }
    s.out[t][2 * (size_t)i] = a;
    s.out[t][2 * (size_t)i + 1] = b;
    if (boxes)
    {
      double ox, oy, oz;
      rigid_apply(T0, (double)a.x, (double)a.y, (double)a.z, ox, oy, oz);
      const float v[3] = {(float)ox, (float)oy, (float)oz};
      for (int d = 0; d < 3; ++d)
        if (v[d] == v[d]) lo[d] = hi[d] = f2ou(v[d]);
    }
  }
  if (!boxes) return;
  for (int d = 0; d < 3; ++d)
    for (int sft = 32; sft > 0; sft >>= 1)
    {
      const unsigned l2 = __shfl_down(lo[d], sft), h2 = __shfl_down(hi[d], sft);
      lo[d] = l2 < lo[d] ? l2 : lo[d];
      hi[d] = h2 > hi[d] ? h2 : hi[d];
    }
  if ((threadIdx.x & 63) == 0)
    for (int d = 0; d < 3; ++d) { slo[threadIdx.x >> 6][d] = lo[d]; shi[threadIdx.x >> 6][d] = hi[d]; }
  __syncthreads();
  if (threadIdx.x < 3)
  {
    const int d = threadIdx.x;
    unsigned l = slo[0][d], h = shi[0][d];
    for (int w = 1; w < 4; ++w) { l = slo[w][d] < l ? slo[w][d] : l; h = shi[w][d] > h ? shi[w][d] : h; }
    atomicMin(&bits[6 * t + d], l);
    atomicMax(&bits[6 * t + 3 + d], h);
  }
}

// Everything of LinearTransformInterpolator that is independent of the point (lsa_posemath.h: one source for the host
// and for the device, which prepares the same constants behind a solve)
InterpConst make_interp(const double H0[16], const double H1[16], double t0, double t1)
{
  posemath::Pose a, b;
  std::memcpy(a.m, H0, sizeof(a.m));
  std::memcpy(b.m, H1, sizeof(b.m));
  return posemath::MakeInterpConst(a, b, t0, t1);
}

}  // namespace

namespace lsa
{
InterpConst make_interp_const(const double H0[16], const double H1[16], double t0, double t1) { return make_interp(H0, H1, t0, t1); }

// time range of a keypoint set into range_bits[0..1]; counts_dev: device counts (extraction, host counts not known yet) or nullptr
int enqueue_time_range(lsa_ctx* ctx, int set, const int* counts_dev)
{
  KpSets sets;
  int nmax = 0;
  for (int k = 0; k < 3; ++k)
  {
    sets.pts[k] = reinterpret_cast<const float4*>(ctx->kp[set][k]);
    sets.n[k] = ctx->kp_n[set][k];
    nmax = std::max(nmax, sets.n[k]);
  }
  if (counts_dev) nmax = ctx->frame_n;  // upper bound of every count
  hipLaunchKernelGGL(k_range_init, dim3(1), dim3(64), 0, ctx->stream, ctx->range_bits);
  if (nmax > 0)
    hipLaunchKernelGGL(k_time_range3, dim3((nmax + 255) / 256, 3), dim3(256), 0, ctx->stream, sets, counts_dev, ctx->range_bits);
  return LSA_OK;
}
void finish_time_range(lsa_ctx* ctx, int set, const unsigned long long bits[2])
{
  const int total = ctx->kp_n[set][0] + ctx->kp_n[set][1] + ctx->kp_n[set][2];
  // empty set: the reference's min/max loop leaves its initial values (Slam.cxx:1291-1300)
  ctx->kp_time[set][0] = total == 0 ? std::numeric_limits<double>::max() : o2d_host(bits[0]);
  ctx->kp_time[set][1] = total == 0 ? std::numeric_limits<double>::lowest() : o2d_host(bits[1]);
  ctx->kp_time_valid[set] = true;
}
}  // namespace lsa

extern "C" {

int lsa_reset_working_keypoints(lsa_ctx* ctx)
{
  if (!ctx) return LSA_E_ARG;
  LSA_HIP(ctx, hipSetDevice(ctx->device));
  StageOut so;
  int nmax = 0;
  for (int k = 0; k < 3; ++k)
  {
    const int n = ctx->kp_n[LSA_SET_RAW_CURRENT][k];
    ctx->kp_n[LSA_SET_WORKING][k] = n;
    so.in[k] = reinterpret_cast<const float4*>(ctx->kp[LSA_SET_RAW_CURRENT][k]);
    so.out[k] = reinterpret_cast<float4*>(ctx->kp[LSA_SET_WORKING][k]);
    so.n[k] = n;
    nmax = std::max(nmax, n);
  }
  if (nmax > 0) hipLaunchKernelGGL(k_copy_sets, dim3((nmax + 255) / 256, 3), dim3(256), 0, ctx->stream, so);  // one launch instead of three copies
  ctx->loc_boxes = false;
  ctx->kp_time_valid[LSA_SET_WORKING] = ctx->kp_time_valid[LSA_SET_RAW_CURRENT];
  ctx->kp_time[LSA_SET_WORKING][0] = ctx->kp_time[LSA_SET_RAW_CURRENT][0];
  ctx->kp_time[LSA_SET_WORKING][1] = ctx->kp_time[LSA_SET_RAW_CURRENT][1];
  return LSA_OK;
}

int lsa_undistort(lsa_ctx* ctx, const double H0[16], const double H1[16], double t0, double t1)
{
  if (!ctx || !H0 || !H1) return ctx ? ctx->fail(LSA_E_ARG, "lsa_undistort: bad argument") : LSA_E_ARG;
  LSA_HIP(ctx, hipSetDevice(ctx->device));
  const InterpConst c = make_interp(H0, H1, t0, t1);
  KpSetsRW sets;
  int nmax = 0, total = 0;
  for (int k = 0; k < 3; ++k)
  {
    sets.pts[k] = reinterpret_cast<float4*>(ctx->kp[LSA_SET_WORKING][k]);
    sets.n[k] = ctx->kp_n[LSA_SET_WORKING][k];
    nmax = std::max(nmax, sets.n[k]);
    total += std::max(sets.n[k], 0);
  }
  if (nmax <= 0) return LSA_OK;
  ProfScope ps(ctx, "undistort", (double)total * 48);
  hipLaunchKernelGGL(k_undistort, dim3((nmax + 255) / 256, 3), dim3(256), 0, ctx->stream, sets, c);
  return LSA_OK;
}

int lsa_arm_localization_boxes(lsa_ctx* ctx)
{
  if (!ctx) return LSA_E_ARG;
  if (ctx->loc_armed) return LSA_OK;
  LSA_HIP(ctx, hipSetDevice(ctx->device));
  hipLaunchKernelGGL(k_bbox_init, dim3(1), dim3(64), 0, ctx->stream, lsa::loc_box_words(ctx));
  ctx->loc_armed = true;
  return LSA_OK;
}

int lsa_localization_begin(lsa_ctx* ctx, const double H0[16], const double H1[16], double t0, double t1, const double box_pose[16])
{
  if (!ctx || (H0 && !H1)) return ctx ? ctx->fail(LSA_E_ARG, "lsa_localization_begin: bad argument") : LSA_E_ARG;
  LSA_HIP(ctx, hipSetDevice(ctx->device));
  StageOut so;
  int nmax = 0, total = 0;
  for (int k = 0; k < 3; ++k)
  {
    const int n = ctx->kp_n[LSA_SET_RAW_CURRENT][k];
    ctx->kp_n[LSA_SET_WORKING][k] = n;
    so.in[k] = reinterpret_cast<const float4*>(ctx->kp[LSA_SET_RAW_CURRENT][k]);
    so.out[k] = reinterpret_cast<float4*>(ctx->kp[LSA_SET_WORKING][k]);
    so.n[k] = n;
    if (box_pose) ctx->bbox_n[k] = n;
    nmax = std::max(nmax, n);
    total += std::max(n, 0);
  }
  ctx->kp_time_valid[LSA_SET_WORKING] = ctx->kp_time_valid[LSA_SET_RAW_CURRENT];
  ctx->kp_time[LSA_SET_WORKING][0] = ctx->kp_time[LSA_SET_RAW_CURRENT][0];
  ctx->kp_time[LSA_SET_WORKING][1] = ctx->kp_time[LSA_SET_RAW_CURRENT][1];
  ctx->loc_boxes = box_pose != nullptr;
  if (box_pose)
  {
    // the words are armed while the device is idle between two frames (lsa_arm_localization_boxes); here only when nobody did
    const int rc = lsa_arm_localization_boxes(ctx);
    if (rc) return rc;
    ctx->loc_armed = false;
    ctx->bbox_pending = true;
    ctx->bbox_copied = false;
  }
  if (nmax <= 0) return LSA_OK;
  InterpConst c{};
  if (H0) c = make_interp(H0, H1, t0, t1);
  Rigid T{};
  if (box_pose) row_major_to_rt(box_pose, T.R, T.t);
  ProfScope ps(ctx, "undistort", (double)total * (H0 ? 64 : 64));
  hipLaunchKernelGGL(k_loc_start, dim3((nmax + 255) / 256, 3), dim3(256), 0, ctx->stream, so, H0 ? 1 : 0, c, box_pose ? 1 : 0, T, lsa::loc_box_words(ctx));
  return LSA_OK;
}

int lsa_working_time_range(lsa_ctx* ctx, double* tmin, double* tmax) { return lsa_keypoint_time_range(ctx, LSA_SET_WORKING, tmin, tmax); }

int lsa_keypoint_time_range(lsa_ctx* ctx, int set, double* tmin, double* tmax)
{
  if (!ctx || !tmin || !tmax || set < 0 || set > 2) return ctx ? ctx->fail(LSA_E_ARG, "lsa_keypoint_time_range: bad argument") : LSA_E_ARG;
  LSA_HIP(ctx, hipSetDevice(ctx->device));
  if (!ctx->kp_time_valid[set])
  {
    // (the usual case costs nothing: lsa_extract_keypoints reduced the range of the raw keypoints and
    // lsa_reset_working_keypoints carried it over)
    int rc = lsa::enqueue_time_range(ctx, set, nullptr);
    if (rc) return rc;
    unsigned long long* hp = reinterpret_cast<unsigned long long*>(ctx->host_pinned + 128);
    LSA_HIP(ctx, hipMemcpyAsync(hp, ctx->range_bits, 2 * sizeof(unsigned long long), hipMemcpyDeviceToHost, ctx->stream));
    LSA_HIP(ctx, hipStreamSynchronize(ctx->stream));
    lsa::finish_time_range(ctx, set, hp);
  }
  *tmin = ctx->kp_time[set][0];
  *tmax = ctx->kp_time[set][1];
  return LSA_OK;
}
static int bboxes_begin(lsa_ctx* ctx, int set, const double pose[16], const double pose_end[16], double t0, double t1)
{
  if (!ctx || !pose || set < 0 || set > 2) return ctx ? ctx->fail(LSA_E_ARG, "lsa_keypoint_bboxes_begin: bad argument") : LSA_E_ARG;
  LSA_HIP(ctx, hipSetDevice(ctx->device));
  KpSets sets;
  int nmax = 0;
  for (int k = 0; k < 3; ++k)
  {
    sets.pts[k] = reinterpret_cast<const float4*>(ctx->kp[set][k]);
    sets.n[k] = ctx->bbox_n[k] = ctx->kp_n[set][k];
    nmax = std::max(nmax, sets.n[k]);
  }
  ctx->bbox_pending = true;
  ctx->pred_on_lookahead = false;
  if (nmax <= 0) return LSA_OK;
  Rigid T;
  row_major_to_rt(pose, T.R, T.t);
  unsigned* bits = reinterpret_cast<unsigned*>(ctx->range_bits + 16);
  hipLaunchKernelGGL(k_bbox_init, dim3(1), dim3(64), 0, ctx->stream, bits);
  InterpConst ic{};
  if (pose_end) ic = make_interp(pose, pose_end, t0, t1);
  hipLaunchKernelGGL(k_bbox3, dim3((nmax + 255) / 256, 3), dim3(256), 0, ctx->stream, sets, T, pose_end ? 1 : 0, ic, bits);
  LSA_HIP(ctx, hipMemcpyAsync(ctx->host_pinned + 160, bits, 18 * sizeof(unsigned), hipMemcpyDeviceToHost, ctx->stream));
  LSA_HIP(ctx, hipEventRecord(ctx->ev_bbox, ctx->stream));
  ctx->bbox_copied = true;
  return LSA_OK;
}
int lsa_keypoint_bboxes_begin(lsa_ctx* ctx, int set, const double pose[16]) { return bboxes_begin(ctx, set, pose, nullptr, 0., 0.); }
int lsa_keypoint_bboxes_begin_interp(lsa_ctx* ctx, int set, const double H0[16], const double H1[16], double t0, double t1)
{
  if (!H1) return ctx ? ctx->fail(LSA_E_ARG, "lsa_keypoint_bboxes_begin_interp: bad argument") : LSA_E_ARG;
  return bboxes_begin(ctx, set, H0, H1, t0, t1);
}
// The boxes of a PREDICTION (the sub-maps extracted ahead of time, lsa_device_grid_submap_ahead_begin): same words, same
// kernels, but on the look-ahead stream -- where the grids that read them work -- and never copied to the host: nothing of it
// touches the context's stream, whose next search would otherwise queue behind it.  _mark (the thread that owns the
// context's stream, once the keypoints are enqueued) says from where on the keypoints exist; the boxes may then be
// enqueued by any thread.
int lsa_keypoint_boxes_predicted_mark(lsa_ctx* ctx)
{
  if (!ctx) return LSA_E_ARG;
  LSA_HIP(ctx, hipSetDevice(ctx->device));
  LSA_HIP(ctx, hipEventRecord(ctx->ev_pred, ctx->stream));
  return LSA_OK;
}
int lsa_keypoint_boxes_predicted(lsa_ctx* ctx, int set, const double H0[16], const double H1[16], double t0, double t1)
{
  if (!ctx || !H0 || set < 0 || set > 2) return ctx ? ctx->fail(LSA_E_ARG, "lsa_keypoint_boxes_predicted: bad argument") : LSA_E_ARG;
  LSA_HIP(ctx, hipSetDevice(ctx->device));
  KpSets sets;
  int nmax = 0;
  for (int k = 0; k < 3; ++k)
  {
    sets.pts[k] = reinterpret_cast<const float4*>(ctx->kp[set][k]);
    sets.n[k] = ctx->kp_n[set][k];
    nmax = std::max(nmax, sets.n[k]);
  }
  if (nmax <= 0) return LSA_OK;
  Rigid T;
  row_major_to_rt(H0, T.R, T.t);
  InterpConst ic{};
  if (H1) ic = make_interp(H0, H1, t0, t1);
  hipStream_t st = ctx->prefetch_stream;
  LSA_HIP(ctx, hipStreamWaitEvent(st, ctx->ev_pred, 0));
  unsigned* bits = lsa::box_words(ctx);
  hipLaunchKernelGGL(k_bbox_init, dim3(1), dim3(64), 0, st, bits);
  hipLaunchKernelGGL(k_bbox3, dim3((nmax + 255) / 256, 3), dim3(256), 0, st, sets, T, H1 ? 1 : 0, ic, bits);
  ctx->pred_on_lookahead = true;
  return LSA_OK;
}

int lsa_keypoint_bboxes_end(lsa_ctx* ctx, float mn[9], float mx[9])
{
  if (!ctx || !mn || !mx) return ctx ? ctx->fail(LSA_E_ARG, "lsa_keypoint_bboxes_end: bad argument") : LSA_E_ARG;
  if (!ctx->bbox_pending) return ctx->fail(LSA_E_STATE, "lsa_keypoint_bboxes_end: no lsa_keypoint_bboxes_begin before");
  ctx->bbox_pending = false;
  for (int i = 0; i < 9; ++i) { mn[i] = FLT_MAX; mx[i] = -FLT_MAX; }
  if (ctx->bbox_n[0] <= 0 && ctx->bbox_n[1] <= 0 && ctx->bbox_n[2] <= 0) return LSA_OK;
  if (!ctx->bbox_copied)
  {
    // the boxes lsa_localization_begin left on the device: nobody needs them on the host in the pipeline, so they only
    // come over when they are asked for
    LSA_HIP(ctx, hipMemcpyAsync(ctx->host_pinned + 160, lsa::loc_box_words(ctx), 18 * sizeof(unsigned), hipMemcpyDeviceToHost, ctx->stream));
    LSA_HIP(ctx, hipEventRecord(ctx->ev_bbox, ctx->stream));
    ctx->bbox_copied = true;
  }
  LSA_HIP(ctx, hipEventSynchronize(ctx->ev_bbox));
  const unsigned* hp = reinterpret_cast<const unsigned*>(ctx->host_pinned + 160);
  for (int k = 0; k < 3; ++k)
    if (ctx->bbox_n[k] > 0)
      for (int d = 0; d < 3; ++d) { mn[3 * k + d] = ou2f_host(hp[6 * k + d]); mx[3 * k + d] = ou2f_host(hp[6 * k + 3 + d]); }
  return LSA_OK;
}
int lsa_working_bboxes(lsa_ctx* ctx, const double pose[16], float mn[9], float mx[9])
{
  if (!ctx || !pose || !mn || !mx) return ctx ? ctx->fail(LSA_E_ARG, "lsa_working_bboxes: bad argument") : LSA_E_ARG;
  const int rc = lsa_keypoint_bboxes_begin(ctx, LSA_SET_WORKING, pose);
  return rc ? rc : lsa_keypoint_bboxes_end(ctx, mn, mx);
}
int lsa_working_bbox(lsa_ctx* ctx, int type, const double pose[16], float mn[3], float mx[3])
{
  if (!ctx || !pose || type < 0 || type > 2) return ctx ? ctx->fail(LSA_E_ARG, "lsa_working_bbox: bad argument") : LSA_E_ARG;
  float lo[9], hi[9];
  const int rc = lsa_working_bboxes(ctx, pose, lo, hi);
  if (rc) return rc;
  for (int d = 0; d < 3; ++d) { mn[d] = lo[3 * type + d]; mx[d] = hi[3 * type + d]; }
  return LSA_OK;
}

int lsa_stage_transformed(lsa_ctx* ctx, int set, const double pose[16])
{
  if (!ctx || !pose || set < 0 || set > 2) return ctx ? ctx->fail(LSA_E_ARG, "lsa_stage_transformed: bad argument") : LSA_E_ARG;
  LSA_HIP(ctx, hipSetDevice(ctx->device));
  StageOut so;
  int nmax = 0;
  for (int k = 0; k < 3; ++k)
  {
    const int n = ctx->kp_n[set][k];
    if (n > ctx->stage_cap[k])
    {
      retire_host(ctx, ctx->stage[k]);
      ctx->stage[k] = nullptr;
      const int cap = std::max(n + n / 4, 4096);
      LSA_HIP(ctx, hipHostMalloc((void**)&ctx->stage[k], (size_t)cap * sizeof(lsa_point_t), hipHostMallocDefault));
      ctx->stage_cap[k] = cap;
    }
    so.in[k] = reinterpret_cast<const float4*>(ctx->kp[set][k]);
    so.out[k] = reinterpret_cast<float4*>(ctx->stage[k]);
    so.n[k] = ctx->stage_n[k] = n;
    nmax = std::max(nmax, n);
  }
  if (nmax > 0)
  {
    Rigid T;
    row_major_to_rt(pose, T.R, T.t);
    ProfScope ps(ctx, "transform_keypoints_out", (double)(so.n[0] + so.n[1] + so.n[2]) * 64);
    hipLaunchKernelGGL(k_transform_stage, dim3((nmax + 255) / 256, 3), dim3(256), 0, ctx->stream, so, T);
  }
  LSA_HIP(ctx, hipEventRecord(ctx->ev_stage, ctx->stream));
  ctx->stage_pending = true;
  return LSA_OK;
}
}  // extern "C"

namespace lsa
{
// src (n points on the device) moved by `pose` into dst (device), on `stream` (the context's when null)
int transform_points_to(lsa_ctx* ctx, const lsa_point_t* src, int n, const double pose[16], lsa_point_t* dst, hipStream_t stream)
{
  if (n <= 0) return LSA_OK;
  StageOut so{};
  so.in[0] = reinterpret_cast<const float4*>(src);
  so.out[0] = reinterpret_cast<float4*>(dst);
  so.n[0] = n;
  Rigid T;
  row_major_to_rt(pose, T.R, T.t);
  hipLaunchKernelGGL(k_transform_stage, dim3((n + 255) / 256, 1), dim3(256), 0, stream ? stream : ctx->stream, so, T);
  return LSA_OK;
}
// ... up to three sets at once (one launch, a block row per set; n <= 0: nothing for that row)
int transform_sets_to(lsa_ctx* ctx, const lsa_point_t* const src[3], const int n[3], const double pose[16], lsa_point_t* const dst[3], hipStream_t stream)
{
  StageOut so{};
  int nmax = 0;
  for (int k = 0; k < 3; ++k)
  {
    so.in[k] = reinterpret_cast<const float4*>(src[k]);
    so.out[k] = reinterpret_cast<float4*>(dst[k]);
    so.n[k] = n[k] > 0 ? n[k] : 0;
    nmax = std::max(nmax, so.n[k]);
  }
  if (nmax <= 0) return LSA_OK;
  Rigid T;
  row_major_to_rt(pose, T.R, T.t);
  hipLaunchKernelGGL(k_transform_stage, dim3((nmax + 255) / 256, 3), dim3(256), 0, stream ? stream : ctx->stream, so, T);
  return LSA_OK;
}
}  // namespace lsa

extern "C" {

int lsa_staged_transformed(lsa_ctx* ctx, int type, const lsa_point_t** pts, int* n)
{
  if (!ctx || type < 0 || type > 2 || !pts || !n) return LSA_E_ARG;
  if (!ctx->stage_pending) return LSA_E_STATE;
  // may be called from another host thread than the one that staged
  if (hipSetDevice(ctx->device) != hipSuccess || hipEventSynchronize(ctx->ev_stage) != hipSuccess) return LSA_E_HIP;
  *pts = ctx->stage[type];
  *n = ctx->stage_n[type];
  return LSA_OK;
}

int lsa_download_transformed(lsa_ctx* ctx, int set, int type, const double pose[16], lsa_point_t* out, int capacity)
{
  if (!ctx || !pose || set < 0 || set > 2 || type < 0 || type > 2) return ctx ? ctx->fail(LSA_E_ARG, "lsa_download_transformed: bad argument") : LSA_E_ARG;
  LSA_HIP(ctx, hipSetDevice(ctx->device));
  const int n = std::min(capacity, ctx->kp_n[set][type]);
  if (n <= 0) return 0;
  int rc = ensure_scratch(ctx, (size_t)n * sizeof(lsa_point_t));
  if (rc) return rc;
  Rigid T;
  row_major_to_rt(pose, T.R, T.t);
  InterpConst dummy{};
  {
    ProfScope ps(ctx, "transform_keypoints_out", (double)n * 64);
    hipLaunchKernelGGL(k_transform_out, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, reinterpret_cast<const float4*>(ctx->kp[set][type]), n, 0, dummy,
                       T, 0., reinterpret_cast<float4*>(ctx->scratch_out));
  }
  LSA_HIP(ctx, hipMemcpyAsync(out, ctx->scratch_out, (size_t)n * sizeof(lsa_point_t), hipMemcpyDeviceToHost, ctx->stream));
  LSA_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return n;
}

int lsa_transform_frame(lsa_ctx* ctx, int interpolate, const double H0[16], const double H1[16], double t0, double t1, lsa_point_t* out,
                        int capacity)
{
  return lsa_transform_frame_at(ctx, interpolate, H0, H1, t0, t1, 0., out, capacity);
}

int lsa_transform_frame_at(lsa_ctx* ctx, int interpolate, const double H0[16], const double H1[16], double t0, double t1, double time_offset,
                           lsa_point_t* out, int capacity)
{
  if (!ctx || !H0 || (interpolate && !H1) || !out) return ctx ? ctx->fail(LSA_E_ARG, "lsa_transform_frame: bad argument") : LSA_E_ARG;
  if (!ctx->frame || ctx->frame_n <= 0) return ctx->fail(LSA_E_STATE, "lsa_transform_frame: no frame");
  LSA_HIP(ctx, hipSetDevice(ctx->device));
  const int n = ctx->frame_n;
  if (capacity < n) return ctx->fail(LSA_E_CAPACITY, "lsa_transform_frame: capacity < frame size");
  int rc = ensure_scratch(ctx, (size_t)n * sizeof(lsa_point_t));
  if (rc) return rc;
  Rigid T;
  row_major_to_rt(H0, T.R, T.t);
  InterpConst c{};
  if (interpolate) c = make_interp(H0, H1, t0, t1);
  {
    ProfScope ps(ctx, "transform_frame", (double)n * 64);
    hipLaunchKernelGGL(k_transform_out, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, reinterpret_cast<const float4*>(ctx->frame), n, interpolate, c, T,
                       time_offset, reinterpret_cast<float4*>(ctx->scratch_out));
  }
  LSA_HIP(ctx, hipMemcpyAsync(out, ctx->scratch_out, (size_t)n * sizeof(lsa_point_t), hipMemcpyDeviceToHost, ctx->stream));
  LSA_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return n;
}

}  // extern "C"
