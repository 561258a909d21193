// lsa_selftest.hip -- lsa_selftest_math: evaluates on the device the elementary operations the
// bit-exact CPU/GPU parity rests on (portable sin/cos/atan2/asin/acos of include/lsa_pmath.h, IEEE sqrt and
// division in float and double), so that a test can compare them bit for bit with the host.
#include "lsa_ctx.h"
#include "lsa_device_math.h"

using namespace lsa;

namespace
{
__global__ void k_selftest(int fn, const double* __restrict__ x, const double* __restrict__ y, int n, double* __restrict__ out)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double r = 0.;
  switch (fn)
  {
    case 0: r = lsa_sin(x[i]); break;
    case 1: r = lsa_cos(x[i]); break;
    case 2: r = lsa_atan2(y[i], x[i]); break;
    case 3: r = (double)sqrt_t((float)x[i]); break;
    case 4: r = (double)((float)x[i] / (float)y[i]); break;
    case 5: r = sqrt_t(x[i]); break;
    case 6: r = x[i] / y[i]; break;
    case 7: r = lsa_asin(x[i]); break;
    case 8: r = lsa_acos(x[i]); break;
  }
  out[i] = r;
}
}  // namespace

extern "C" int lsa_selftest_math(lsa_ctx* ctx, int fn, const double* x, const double* y, int n, double* out)
{
  if (!ctx || !x || !y || !out || n <= 0 || fn < 0 || fn > 8) return ctx ? ctx->fail(LSA_E_ARG, "lsa_selftest_math: bad argument") : LSA_E_ARG;
  LSA_HIP(ctx, hipSetDevice(ctx->device));
  int rc = ensure_scratch(ctx, (size_t)n * 3 * sizeof(double));
  if (rc) return rc;
  double* dx = (double*)ctx->scratch_out;
  double* dy = dx + n;
  double* dout = dy + n;
  LSA_HIP(ctx, hipMemcpyAsync(dx, x, (size_t)n * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  LSA_HIP(ctx, hipMemcpyAsync(dy, y, (size_t)n * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  hipLaunchKernelGGL(k_selftest, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, fn, dx, dy, n, dout);
  LSA_HIP(ctx, hipMemcpyAsync(out, dout, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  LSA_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return LSA_OK;
}

// Diagnostic (scripts/clock_probe.py): `blocks` single-wave workgroups that sleep-spin for `ms` milliseconds of the
// constant 100 MHz wall clock on a stream of their own and then exit -- a load that occupies next to nothing, to see what the
// clock management of a mostly idle GPU does to the latency of the pipeline's short kernels.
__global__ void k_keep_busy(unsigned long long ticks)
{
  const unsigned long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
}
extern "C" int lsa_selftest_keep_busy(lsa_ctx* ctx, int ms, int blocks)
{
  if (!ctx || ms < 0 || ms > 2000 || blocks < 1 || blocks > 256) return ctx ? ctx->fail(LSA_E_ARG, "lsa_selftest_keep_busy: bad argument") : LSA_E_ARG;
  LSA_HIP(ctx, hipSetDevice(ctx->device));
  static hipStream_t side = nullptr;
  if (!side) LSA_HIP(ctx, hipStreamCreateWithFlags(&side, hipStreamNonBlocking));
  hipLaunchKernelGGL(k_keep_busy, dim3(blocks), dim3(64), 0, side, (unsigned long long)ms * 100000ull);
  return LSA_OK;
}
