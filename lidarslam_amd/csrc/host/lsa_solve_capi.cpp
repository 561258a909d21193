// lsa_solve_capi.cpp -- C ABI of seam 3 (LocalOptimizer): lsa_solve, lsa_registration_error.
#include <cstring>
#include "lsa_lm.h"

using namespace lsa::host;

extern "C" {

int lsa_solve(lsa_ctx* ctx, unsigned type_mask, const double prior[16], int lm_max_iter, int two_d_mode, double optimized[16], int summary[4],
              double costs[2])
{
  if (!ctx || !prior || !optimized) return ctx ? ctx->fail(LSA_E_ARG, "lsa_solve: bad argument") : LSA_E_ARG;
  Pose p;
  std::memcpy(p.m, prior, sizeof(p.m));
  LocalOptimizer opt(ctx);
  opt.SetDeviceLoop(false);  // this entry point is the host-driven loop; lsa_solve_device is the one-launch solve
  opt.SetTwoDMode(two_d_mode != 0);
  opt.SetLMMaxIter(lm_max_iter < 0 ? 0 : (unsigned)lm_max_iter);
  opt.SetPosePrior(p);
  opt.UseDeviceResiduals(type_mask);
  SolveSummary s;
  const int rc = opt.Solve(s);
  if (rc < 0) return rc;
  const Pose o = opt.GetOptimizedPose();
  std::memcpy(optimized, o.m, sizeof(o.m));
  if (summary)
  {
    summary[0] = s.num_successful_steps; summary[1] = s.num_unsuccessful_steps;
    summary[2] = s.num_iterations; summary[3] = s.num_evaluations;
  }
  if (costs) { costs[0] = s.initial_cost; costs[1] = s.final_cost; }
  return LSA_OK;
}

int lsa_registration_error(lsa_ctx* ctx, unsigned type_mask, const double pose[16], int two_d_mode, double cov[36], double err[2])
{
  if (!ctx || !pose || !cov) return ctx ? ctx->fail(LSA_E_ARG, "lsa_registration_error: bad argument") : LSA_E_ARG;
  Pose p;
  std::memcpy(p.m, pose, sizeof(p.m));
  LocalOptimizer opt(ctx);
  opt.SetTwoDMode(two_d_mode != 0);
  opt.SetPosePrior(p);
  opt.UseDeviceResiduals(type_mask);
  RegistrationError e;
  const int rc = opt.EstimateRegistrationError(e);
  if (rc < 0) return rc;
  std::memcpy(cov, e.Covariance.data(), 36 * sizeof(double));
  if (err) { err[0] = e.PositionError; err[1] = e.OrientationError; }
  return LSA_OK;
}

}  // extern "C"
