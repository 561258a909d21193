// lsa_lm.cpp -- see lsa_lm.h.  Solver options: slam_lib/src/LocalOptimizer.cxx:93-96
// (DENSE_QR, max_num_iterations = LMMaxIter, everything else Ceres defaults).
#include "lsa_lm.h"
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace lsa
{
namespace host
{
namespace
{

struct Eval
{
  double cost = 0.;
  double g[6] = {0};
  double H[36] = {0};
  int nValid = 0;
};

// small dense symmetric positive definite solve (normal equations of the augmented LM system)
bool SolveSPD(int n, const double* A, const double* b, double* x)
{
  // (divisions by a diagonal element are multiplications by its reciprocal, taken once: as solve_spd of lsa_lm.hip)
  double L[36], rinv[6];
  for (int i = 0; i < n; ++i)
    for (int j = 0; j <= i; ++j)
    {
      double s = A[i * n + j];
      for (int k = 0; k < j; ++k) s -= L[i * n + k] * L[j * n + k];
      if (i == j)
      {
        if (!(s > 0.0) || !std::isfinite(s)) return false;
        L[i * n + i] = std::sqrt(s);
        rinv[i] = 1.0 / L[i * n + i];
      }
      else
        L[i * n + j] = s * rinv[j];
    }
  double y[6];
  for (int i = 0; i < n; ++i)
  {
    double s = b[i];
    for (int k = 0; k < i; ++k) s -= L[i * n + k] * y[k];
    y[i] = s * rinv[i];
  }
  for (int i = n - 1; i >= 0; --i)
  {
    double s = y[i];
    for (int k = i + 1; k < n; ++k) s -= L[k * n + i] * x[k];
    x[i] = s * rinv[i];
  }
  for (int i = 0; i < n; ++i)
    if (!std::isfinite(x[i])) return false;
  return true;
}

// cyclic Jacobi eigen-decomposition of a small symmetric matrix; ascending eigenvalues
void SymEigen(int n, const double* Ain, double* evals, double* evecs)
{
  double A[36], V[36];
  for (int i = 0; i < n * n; ++i) A[i] = Ain[i];
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < n; ++j) V[i * n + j] = (i == j);
  // squared Frobenius norm: what the off-diagonal part is held against.  (The first version swept until that part was below
  // 1e-300 -- in practice until it had underflowed: twenty microseconds per frame for digits nobody can see.)
  double scale = 0;
  for (int i = 0; i < n * n; ++i) scale += A[i] * A[i];
  for (int sweep = 0; sweep < 100; ++sweep)
  {
    double off = 0;
    for (int i = 0; i < n; ++i)
      for (int j = i + 1; j < n; ++j) off += A[i * n + j] * A[i * n + j];
    if (off <= 1e-36 * scale) break;  // (machine precision is 5e-32 of the squared norm; Jacobi converges quadratically: one more sweep past it)
    for (int p = 0; p < n; ++p)
      for (int q = p + 1; q < n; ++q)
      {
        const double apq = A[p * n + q];
        if (apq == 0.0) continue;
        const double theta = (A[q * n + q] - A[p * n + p]) / (2.0 * apq);
        const double t = (theta >= 0 ? 1.0 : -1.0) / (std::abs(theta) + std::sqrt(theta * theta + 1.0));
        const double c = 1.0 / std::sqrt(t * t + 1.0), s = t * c;
        for (int k = 0; k < n; ++k)
        {
          const double akp = A[k * n + p], akq = A[k * n + q];
          A[k * n + p] = c * akp - s * akq;
          A[k * n + q] = s * akp + c * akq;
        }
        for (int k = 0; k < n; ++k)
        {
          const double apk = A[p * n + k], aqk = A[q * n + k];
          A[p * n + k] = c * apk - s * aqk;
          A[q * n + k] = s * apk + c * aqk;
        }
        for (int k = 0; k < n; ++k)
        {
          const double vkp = V[k * n + p], vkq = V[k * n + q];
          V[k * n + p] = c * vkp - s * vkq;
          V[k * n + q] = s * vkp + c * vkq;
        }
      }
  }
  int order[6];
  for (int i = 0; i < n; ++i) order[i] = i;
  std::sort(order, order + n, [&](int a, int b) { return A[a * n + a] < A[b * n + b]; });
  for (int i = 0; i < n; ++i)
  {
    evals[i] = A[order[i] * n + order[i]];
    for (int k = 0; k < n; ++k) evecs[k * n + i] = V[k * n + order[i]];
  }
}

}  // namespace

void LocalOptimizer::TakeResult(const lsa_solve_result_t& r, SolveSummary& sum)
{
  sum = SolveSummary();
  sum.num_successful_steps = r.num_successful_steps;
  sum.num_unsuccessful_steps = r.num_unsuccessful_steps;
  sum.num_iterations = r.num_iterations;
  sum.num_evaluations = r.num_evaluations;
  sum.initial_cost = r.initial_cost;
  sum.final_cost = r.final_cost;
  sum.num_matches = r.num_matches;
  sum.skipped = r.skipped != 0;
  sum.message = r.message;
  if (!sum.skipped)
  {
    std::memcpy(PoseArray, r.pose, sizeof(PoseArray));
    std::memcpy(FinalH, r.H, sizeof(FinalH));
    HaveFinal = true;
  }
}

int LocalOptimizer::Solve(SolveSummary& sum)
{
  HaveFinal = false;
  if (DeviceLoop)
  {
    lsa_solve_result_t r;
    const int rc = lsa_solve_device(Ctx, TypeMask, PoseArray, TwoDMode ? 1 : 0, static_cast<int>(LMMaxIter), static_cast<int>(MinMatches), &r);
    if (rc == LSA_OK)
    {
      TakeResult(r, sum);
      return LSA_OK;
    }
    if (rc != LSA_E_STATE) return rc;  // LSA_E_STATE: the device gave up, nothing was changed
  }
  return SolveOnHost(sum);
}

int LocalOptimizer::Begin(bool gated)
{
  HaveFinal = false;
  return lsa_solve_device_begin(Ctx, TypeMask, gated ? nullptr : PoseArray, TwoDMode ? 1 : 0, static_cast<int>(LMMaxIter), static_cast<int>(MinMatches));
}

int LocalOptimizer::End(SolveSummary& sum)
{
  HaveFinal = false;
  lsa_solve_result_t r;
  const int rc = lsa_solve_device_end(Ctx, &r);
  if (lsa_icp_trace_on()) std::fprintf(stderr, "[optimizer end] rc %d (%s)\n", rc, rc ? lsa_last_error(Ctx) : "");
  if (rc == LSA_OK)
  {
    TakeResult(r, sum);
    return LSA_OK;
  }
  if (rc != LSA_E_STATE) return rc;  // LSA_E_GATE: neither the match nor the solve ran, the caller redoes both
  // the solve gave up: the residual blocks are there, the loop runs here.  Whatever was enqueued ahead has been called off
  // (lsa_solve_device_end): 1 tells the caller
  const int hrc = SolveOnHost(sum);
  return hrc < 0 ? hrc : 1;
}

int LocalOptimizer::SolveOnHost(SolveSummary& sum)
{
  sum = SolveSummary();
  int act[6], n = 0;
  for (int i = 0; i < 6; ++i)
    if (!(TwoDMode && (i == 2 || i == 3 || i == 4))) act[n++] = i;  // SubsetParameterization(6, {2,3,4}), LocalOptimizer.cxx:89-90

  // ceres::Solver::Options defaults
  const double function_tolerance = 1e-6, gradient_tolerance = 1e-10, parameter_tolerance = 1e-8;
  const double min_relative_decrease = 1e-3, min_trust_region_radius = 1e-32, max_radius = 1e16;
  const double min_diagonal = 1e-6, max_diagonal = 1e32;
  const int max_consecutive_invalid = 5;
  double radius = 1e4, decrease_factor = 2.0;
  bool reuse_diagonal = false;

  auto evaluate = [&](const double* w, bool jac, Eval& e) -> int {
    sum.num_evaluations++;
    return lsa_accumulate(Ctx, TypeMask, w, jac ? 1 : 0, &e.cost, e.g, e.H, &e.nValid);
  };

  double x[6];
  std::memcpy(x, PoseArray, sizeof(x));
  Eval cur;
  int rc = evaluate(x, true, cur);
  if (rc) return rc;
  sum.num_matches = cur.nValid;
  if (static_cast<unsigned>(cur.nValid) < MinMatches)
  {
    sum.skipped = true;
    sum.message = "not enough matches";
    return LSA_OK;
  }
  sum.initial_cost = sum.final_cost = cur.cost;
  sum.num_successful_steps = 1;  // iteration 0 is reported as a successful step by Ceres

  double scale[6];
  for (int a = 0; a < n; ++a) scale[a] = 1.0 / (1.0 + std::sqrt(cur.H[act[a] * 6 + act[a]]));  // Jacobi scaling, once
  auto gradMax = [&](const Eval& e) { double m = 0; for (int a = 0; a < n; ++a) m = std::max(m, std::abs(e.g[act[a]])); return m; };
  auto xnorm = [&](const double* v) { double s = 0; for (int a = 0; a < n; ++a) s += v[act[a]] * v[act[a]]; return std::sqrt(s); };

  if (gradMax(cur) <= gradient_tolerance) { sum.message = "gradient tolerance (iteration 0)"; return LSA_OK; }
  double x_norm = xnorm(x);
  double diag[6] = {0};
  int consecutive_invalid = 0, iter = 0;

  while (true)
  {
    if (iter >= (int)LMMaxIter) { sum.message = "max iterations"; break; }
    if (gradMax(cur) <= gradient_tolerance) { sum.message = "gradient tolerance"; break; }
    if (radius < min_trust_region_radius) { sum.message = "min trust region radius"; break; }
    ++iter;
    sum.num_iterations = iter;

    double Hs[36], gs[6];
    for (int a = 0; a < n; ++a)
    {
      gs[a] = cur.g[act[a]] * scale[a];
      for (int b = 0; b < n; ++b) Hs[a * n + b] = cur.H[act[a] * 6 + act[b]] * scale[a] * scale[b];
    }
    if (!reuse_diagonal)
      for (int a = 0; a < n; ++a) diag[a] = std::min(std::max(Hs[a * n + a], min_diagonal), max_diagonal);
    double M[36], y[6], step[6];
    for (int i = 0; i < n * n; ++i) M[i] = Hs[i];
    for (int a = 0; a < n; ++a) M[a * n + a] += diag[a] / radius;
    bool ok = SolveSPD(n, M, gs, y);
    reuse_diagonal = true;
    double model_cost_change = 0;
    if (ok)
    {
      for (int a = 0; a < n; ++a) step[a] = -y[a];
      double sg = 0, sHs = 0;
      for (int a = 0; a < n; ++a)
      {
        sg += step[a] * gs[a];
        double t = 0;
        for (int b = 0; b < n; ++b) t += Hs[a * n + b] * step[b];
        sHs += step[a] * t;
      }
      model_cost_change = -sg - 0.5 * sHs;
      if (model_cost_change < 0.0) ok = false;
    }
    if (!ok)
    {
      ++sum.num_unsuccessful_steps;
      if (++consecutive_invalid >= max_consecutive_invalid) { sum.message = "too many invalid steps"; break; }
      radius /= decrease_factor; decrease_factor *= 2.0; reuse_diagonal = true;
      continue;
    }
    consecutive_invalid = 0;

    double cand[6], delta_norm = 0;
    std::memcpy(cand, x, sizeof(cand));
    for (int a = 0; a < n; ++a)
    {
      cand[act[a]] = x[act[a]] + step[a] * scale[a];
      const double e = x[act[a]] - cand[act[a]];
      delta_norm += e * e;
    }
    delta_norm = std::sqrt(delta_norm);
    // Ceres evaluates the cost only here and, if the step is accepted, residuals + Jacobian again at
    // the same point.  On the device the Jacobian costs nothing extra (the kernel is bound by the
    // record reads), so it is evaluated speculatively and reused on acceptance: one launch + one
    // read-back per LM iteration instead of two, same arithmetic.
    Eval cc;
    rc = evaluate(cand, true, cc);
    if (rc) return rc;

    // parameter / function tolerance terminate WITHOUT taking the candidate step
    if (delta_norm <= parameter_tolerance * (x_norm + parameter_tolerance)) { sum.message = "parameter tolerance"; break; }
    const double cost_change = cur.cost - cc.cost;
    if (std::abs(cost_change) <= function_tolerance * cur.cost) { sum.message = "function tolerance"; break; }

    const double relative_decrease = cost_change / model_cost_change;
    if (relative_decrease > min_relative_decrease)
    {
      std::memcpy(x, cand, sizeof(x));
      x_norm = xnorm(x);
      cur = cc;  // gradient and J^T J at the accepted point were evaluated speculatively above
      ++sum.num_successful_steps;
      const double t = 2.0 * relative_decrease - 1.0;
      radius = std::min(max_radius, radius / std::max(1.0 / 3.0, 1.0 - t * t * t));
      decrease_factor = 2.0;
      reuse_diagonal = false;
    }
    else
    {
      ++sum.num_unsuccessful_steps;
      radius /= decrease_factor; decrease_factor *= 2.0; reuse_diagonal = true;
    }
  }
  std::memcpy(PoseArray, x, sizeof(x));
  sum.final_cost = cur.cost;
  return LSA_OK;
}

// LocalOptimizer.cxx:112-140: covariance = pseudo-inverse of the loss-corrected J^T J
// (DENSE_SVD, null_space_rank -1 => singular value ratios below sqrt(1e-14) are dropped)
int LocalOptimizer::EstimateRegistrationError(RegistrationError& err)
{
  err = RegistrationError();
  int act[6], n = 0;
  for (int i = 0; i < 6; ++i)
    if (!(TwoDMode && (i == 2 || i == 3 || i == 4))) act[n++] = i;
  Eval cur;
  if (HaveFinal) std::memcpy(cur.H, FinalH, sizeof(FinalH));
  else
  {
    const int rc = lsa_accumulate(Ctx, TypeMask, PoseArray, 1, &cur.cost, cur.g, cur.H, &cur.nValid);
    if (rc) return rc;
  }
  double H[36], evals[6], evecs[36];
  for (int a = 0; a < n; ++a)
    for (int b = 0; b < n; ++b) H[a * n + b] = cur.H[act[a] * 6 + act[b]];
  SymEigen(n, H, evals, evecs);
  const double lmax = evals[n - 1];
  double inv[6];
  bool truncated = false;
  for (int i = n - 1; i >= 0; --i)
  {
    const double ratio = lmax > 0 ? std::sqrt(std::max(evals[i], 0.0) / lmax) : 0.0;
    if (truncated || ratio < std::sqrt(1e-14)) { truncated = true; inv[i] = 0.0; }
    else inv[i] = 1.0 / evals[i];
  }
  for (int a = 0; a < n; ++a)
    for (int b = 0; b < n; ++b)
    {
      double s = 0;
      for (int k = 0; k < n; ++k) s += evecs[a * n + k] * inv[k] * evecs[b * n + k];
      err.Covariance[act[a] * 6 + act[b]] = s;
    }
  auto block = [&](int o, double& e, double* dir) {
    double B[9], ev[3], vec[9];
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) B[i * 3 + j] = err.Covariance[(o + i) * 6 + (o + j)];
    SymEigen(3, B, ev, vec);
    e = std::sqrt(ev[2]);
    for (int i = 0; i < 3; ++i) dir[i] = vec[i * 3 + 2];
  };
  block(0, err.PositionError, err.PositionErrorDirection);
  block(3, err.OrientationError, err.OrientationErrorDirection);
  err.OrientationError = err.OrientationError / M_PI * 180.;
  return LSA_OK;
}

}  // namespace host
}  // namespace lsa
