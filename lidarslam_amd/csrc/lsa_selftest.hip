// lsa_selftest.hip -- lsa_selftest_math: evaluates on the device the elementary operations the
// bit-exact CPU/GPU parity rests on (portable sin/cos/atan2 of include/lsa_pmath.h, IEEE sqrt and
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
  }
  out[i] = r;
}
}  // namespace

extern "C" int lsa_selftest_math(lsa_ctx* ctx, int fn, const double* x, const double* y, int n, double* out)
{
  if (!ctx || !x || !y || !out || n <= 0 || fn < 0 || fn > 6) return ctx ? ctx->fail(LSA_E_ARG, "lsa_selftest_math: bad argument") : LSA_E_ARG;
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
