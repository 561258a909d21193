// ORACLE -- TEST INFRASTRUCTURE ONLY (see orc_math.hpp header).  PARITY UNPINNED.
//
// See orc_lm.hpp.  Residual model: slam_lib/include/LidarSlam/CeresCostFunctions.h:67-79,
// 105-152; loss: slam_lib/src/KeypointsMatcher.cxx:84-101; solver options:
// slam_lib/src/LocalOptimizer.cxx:74-102; covariance: :112-140.
#include "orc_lm.hpp"
#include <cmath>
#include <cstring>
#ifdef _OPENMP
#include <omp.h>
#endif

namespace orc
{
namespace
{
// R = Rz(rz) Ry(ry) Rx(rx) and its partial derivatives (CeresCostFunctions.h:67-79;
// the autodiff Jets of the reference are replaced by the analytic derivative).
struct RotAndDerivs { double R[9], dRx[9], dRy[9], dRz[9]; };
RotAndDerivs MakeRot(double rx, double ry, double rz)
{
  const double cx = t_cos(rx), sx = t_sin(rx);
  const double cy = t_cos(ry), sy = t_sin(ry);
  const double cz = t_cos(rz), sz = t_sin(rz);
  RotAndDerivs o;
  double R[9] = {cy * cz, sx * sy * cz - cx * sz, cx * sy * cz + sx * sz,
                 cy * sz, sx * sy * sz + cx * cz, cx * sy * sz - sx * cz,
                 -sy, sx * cy, cx * cy};
  double dRx[9] = {0, cx * sy * cz + sx * sz, -sx * sy * cz + cx * sz,
                   0, cx * sy * sz - sx * cz, -sx * sy * sz - cx * cz,
                   0, cx * cy, -sx * cy};
  double dRy[9] = {-sy * cz, sx * cy * cz, cx * cy * cz,
                   -sy * sz, sx * cy * sz, cx * cy * sz,
                   -cy, -sx * sy, -cx * sy};
  double dRz[9] = {-cy * sz, -sx * sy * sz - cx * cz, -cx * sy * sz + sx * cz,
                   cy * cz, sx * sy * cz - cx * sz, cx * sy * cz + sx * sz,
                   0, 0, 0};
  std::memcpy(o.R, R, sizeof(R)); std::memcpy(o.dRx, dRx, sizeof(R));
  std::memcpy(o.dRy, dRy, sizeof(R)); std::memcpy(o.dRz, dRz, sizeof(R));
  return o;
}
inline void mv(const double M[9], const double v[3], double o[3])
{
  o[0] = (M[0] * v[0] + M[1] * v[1]) + M[2] * v[2];
  o[1] = (M[3] * v[0] + M[4] * v[1]) + M[5] * v[2];
  o[2] = (M[6] * v[0] + M[7] * v[1]) + M[8] * v[2];
}

// one residual block: r = A (R X + t - P); rho = weight * Tukey_sat(|r|^2)
inline void AccumulateOne(const Residual& res, const double w[6], const RotAndDerivs& rot, bool jac, NormalEq& out)
{
  double RX[3], d[3], r[3];
  mv(rot.R, res.X, RX);
  for (int i = 0; i < 3; ++i) d[i] = (RX[i] + w[i]) - res.P[i];
  mv(res.A, d, r);
  const double s = (r[0] * r[0] + r[1] * r[1]) + r[2] * r[2];
  // ceres::TukeyLoss::Evaluate (Ceres >= 2.0), then ceres::ScaledLoss
  const double a2 = res.sat * res.sat;
  double rho0, rho1;
  if (s <= a2)
  {
    const double value = 1.0 - s / a2;
    const double value_sq = value * value;
    rho0 = a2 / 3.0 * (1.0 - value_sq * value);
    rho1 = value_sq;
  }
  else
  {
    rho0 = a2 / 3.0;
    rho1 = 0.0;
  }
  rho0 *= res.weight;
  rho1 *= res.weight;
  out.cost += 0.5 * rho0;
  out.nValid++;
  if (!jac) return;
  // rho'' <= 0 for Tukey, so the Ceres corrector only scales r and J by sqrt(rho')
  // J = [A | A dRx X | A dRy X | A dRz X]
  double J[3][6];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) J[i][j] = res.A[i * 3 + j];
  double v[3], c[3];
  mv(rot.dRx, res.X, v); mv(res.A, v, c); for (int i = 0; i < 3; ++i) J[i][3] = c[i];
  mv(rot.dRy, res.X, v); mv(res.A, v, c); for (int i = 0; i < 3; ++i) J[i][4] = c[i];
  mv(rot.dRz, res.X, v); mv(res.A, v, c); for (int i = 0; i < 3; ++i) J[i][5] = c[i];
  for (int a = 0; a < 6; ++a)
  {
    out.g[a] += rho1 * ((J[0][a] * r[0] + J[1][a] * r[1]) + J[2][a] * r[2]);
    for (int b = a; b < 6; ++b)
      out.H[a * 6 + b] += rho1 * ((J[0][a] * J[0][b] + J[1][a] * J[1][b]) + J[2][a] * J[2][b]);
  }
}
}  // namespace

void EvaluateResiduals(const Residual* res, size_t n, const double w[6], bool jacobians, NormalEq& out, int nbThreads)
{
  const RotAndDerivs rot = MakeRot(w[3], w[4], w[5]);
  NormalEq acc;
#ifdef _OPENMP
  if (nbThreads > 1)
  {
    std::vector<NormalEq> part(nbThreads);
    #pragma omp parallel num_threads(nbThreads)
    {
      int t = omp_get_thread_num(), nt = omp_get_num_threads();
      size_t b = n * t / nt, e = n * (t + 1) / nt;
      NormalEq local;
      for (size_t i = b; i < e; ++i)
        if (res[i].valid) AccumulateOne(res[i], w, rot, jacobians, local);
      part[t] = local;
    }
    for (int t = 0; t < nbThreads; ++t)
    {
      acc.cost += part[t].cost; acc.nValid += part[t].nValid;
      for (int i = 0; i < 6; ++i) acc.g[i] += part[t].g[i];
      for (int i = 0; i < 36; ++i) acc.H[i] += part[t].H[i];
    }
  }
  else
#endif
  {
    (void)nbThreads;
    for (size_t i = 0; i < n; ++i)
      if (res[i].valid) AccumulateOne(res[i], w, rot, jacobians, acc);
  }
  for (int a = 0; a < 6; ++a)
    for (int b = 0; b < a; ++b) acc.H[a * 6 + b] = acc.H[b * 6 + a];
  out.cost += acc.cost; out.nValid += acc.nValid;
  for (int i = 0; i < 6; ++i) out.g[i] += acc.g[i];
  for (int i = 0; i < 36; ++i) out.H[i] += acc.H[i];
}

bool CholeskySolve(int n, const double* A, const double* b, double* x)
{
  // L L^T = A, forward and backward substitution; every division by a diagonal element is a multiplication by its
  // reciprocal, taken once (the product's device and host loops do the same: operation for operation)
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
  for (int i = 0; i < n; ++i) if (!std::isfinite(x[i])) return false;
  return true;
}

// cyclic Jacobi for small symmetric matrices
void JacobiEigen(int n, const double* Ain, double* evals, double* evecs)
{
  double A[36], V[36];
  for (int i = 0; i < n * n; ++i) A[i] = Ain[i];
  for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) V[i * n + j] = (i == j);
  // squared Frobenius norm: what the off-diagonal part is held against.  (The first version swept until that part was below
  // 1e-300 -- in practice until it had underflowed: twenty microseconds per frame for digits nobody can see.)
  double scale = 0;
  for (int i = 0; i < n * n; ++i) scale += A[i] * A[i];
  for (int sweep = 0; sweep < 100; ++sweep)
  {
    double off = 0;
    for (int i = 0; i < n; ++i) for (int j = i + 1; j < n; ++j) off += A[i * n + j] * A[i * n + j];
    if (off <= 1e-36 * scale) break;  // (machine precision is 5e-32 of the squared norm; Jacobi converges quadratically: one more sweep past it)
    for (int p = 0; p < n; ++p)
      for (int q = p + 1; q < n; ++q)
      {
        double apq = A[p * n + q];
        if (apq == 0.0) continue;
        double theta = (A[q * n + q] - A[p * n + p]) / (2.0 * apq);
        double t = (theta >= 0 ? 1.0 : -1.0) / (std::abs(theta) + std::sqrt(theta * theta + 1.0));
        double c = 1.0 / std::sqrt(t * t + 1.0), s = t * c;
        for (int k = 0; k < n; ++k)
        {
          double akp = A[k * n + p], akq = A[k * n + q];
          A[k * n + p] = c * akp - s * akq;
          A[k * n + q] = s * akp + c * akq;
        }
        for (int k = 0; k < n; ++k)
        {
          double apk = A[p * n + k], aqk = A[q * n + k];
          A[p * n + k] = c * apk - s * aqk;
          A[q * n + k] = s * apk + c * aqk;
        }
        for (int k = 0; k < n; ++k)
        {
          double vkp = V[k * n + p], vkq = V[k * n + q];
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

// ceres::Solve with Solver::Options defaults except linear_solver_type,
// max_num_iterations, num_threads (LocalOptimizer.cxx:93-96).
LMSummary LocalOptimizer::Solve()
{
  LMSummary sum;
  // active parameters (SubsetParameterization(6, {2,3,4}) in 2D mode, LocalOptimizer.cxx:89-90)
  int act[6], n = 0;
  for (int i = 0; i < 6; ++i)
    if (!(TwoDMode && (i == 2 || i == 3 || i == 4))) act[n++] = i;

  // Ceres defaults (trust_region_minimizer / levenberg_marquardt_strategy)
  const double function_tolerance = 1e-6, gradient_tolerance = 1e-10, parameter_tolerance = 1e-8;
  const double min_relative_decrease = 1e-3, min_trust_region_radius = 1e-32, max_radius = 1e16;
  const double min_diagonal = 1e-6, max_diagonal = 1e32;
  const int max_consecutive_invalid = 5;
  double radius = 1e4, decrease_factor = 2.0;
  bool reuse_diagonal = false;

  double x[6];
  std::memcpy(x, PoseArray, sizeof(x));
  NormalEq cur;
  EvaluateResiduals(Residuals.data(), Residuals.size(), x, true, cur, NbThreads);
  sum.num_evaluations++;
  sum.initial_cost = sum.final_cost = cur.cost;
  sum.num_successful_steps = 1;  // iteration 0 counts as successful

  // Jacobi scaling, computed once at iteration 0
  double scale[6];
  for (int a = 0; a < n; ++a) scale[a] = 1.0 / (1.0 + std::sqrt(cur.H[act[a] * 6 + act[a]]));

  auto gradMax = [&](const NormalEq& e) { double m = 0; for (int a = 0; a < n; ++a) m = std::max(m, std::abs(e.g[act[a]])); return m; };
  auto xnorm = [&](const double* v) { double s = 0; for (int a = 0; a < n; ++a) s += v[act[a]] * v[act[a]]; return std::sqrt(s); };

  if (gradMax(cur) <= gradient_tolerance) { sum.message = "gradient tolerance (iteration 0)"; return sum; }
  double x_norm = xnorm(x);
  double diag[6] = {0};
  int consecutive_invalid = 0;
  int iter = 0;

  while (true)
  {
    // FinalizeIterationAndCheckIfMinimizerCanContinue
    if (iter >= (int)LMMaxIter) { sum.message = "max iterations"; break; }
    if (gradMax(cur) <= gradient_tolerance) { sum.message = "gradient tolerance"; break; }
    if (radius < min_trust_region_radius) { sum.message = "min trust region radius"; break; }
    ++iter;
    sum.num_iterations = iter;

    // LevenbergMarquardtStrategy::ComputeStep on the scaled system
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
    bool ok = CholeskySolve(n, M, gs, y);
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
      // HandleInvalidStep
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
      double d = step[a] * scale[a];
      cand[act[a]] = x[act[a]] + d;
      double e = x[act[a]] - cand[act[a]];
      delta_norm += e * e;
    }
    delta_norm = std::sqrt(delta_norm);
    NormalEq cc;
    EvaluateResiduals(Residuals.data(), Residuals.size(), cand, false, cc, NbThreads);
    sum.num_evaluations++;

    // ParameterToleranceReached / FunctionToleranceReached: terminate WITHOUT taking the step
    if (delta_norm <= parameter_tolerance * (x_norm + parameter_tolerance)) { sum.message = "parameter tolerance"; break; }
    const double cost_change = cur.cost - cc.cost;
    if (std::abs(cost_change) <= function_tolerance * cur.cost) { sum.message = "function tolerance"; break; }

    const double relative_decrease = cost_change / model_cost_change;
    if (relative_decrease > min_relative_decrease)
    {
      // HandleSuccessfulStep
      std::memcpy(x, cand, sizeof(x));
      x_norm = xnorm(x);
      NormalEq next;
      EvaluateResiduals(Residuals.data(), Residuals.size(), x, true, next, NbThreads);
      sum.num_evaluations++;
      cur = next;
      ++sum.num_successful_steps;
      double t = 2.0 * relative_decrease - 1.0;
      radius = radius / std::max(1.0 / 3.0, 1.0 - t * t * t);
      radius = std::min(max_radius, radius);
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
  return sum;
}

// LocalOptimizer.cxx:112-140: ceres::Covariance DENSE_SVD, apply_loss_function,
// null_space_rank = -1 (automatic truncation at sqrt(1e-14) singular value ratio)
RegistrationError LocalOptimizer::EstimateRegistrationError()
{
  RegistrationError err;
  int act[6], n = 0;
  for (int i = 0; i < 6; ++i)
    if (!(TwoDMode && (i == 2 || i == 3 || i == 4))) act[n++] = i;
  NormalEq cur;
  EvaluateResiduals(Residuals.data(), Residuals.size(), PoseArray, true, cur, NbThreads);
  double H[36], evals[6], evecs[36];
  for (int a = 0; a < n; ++a) for (int b = 0; b < n; ++b) H[a * n + b] = cur.H[act[a] * 6 + act[b]];
  JacobiEigen(n, H, evals, evecs);
  const double lmax = evals[n - 1];
  double inv[6];
  // singular values of J are sqrt(evals); descending order = from the back
  bool truncated = false;
  for (int i = n - 1; i >= 0; --i)
  {
    double ratio = lmax > 0 ? std::sqrt(std::max(evals[i], 0.0) / lmax) : 0.0;
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
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) B[i * 3 + j] = err.Covariance[(o + i) * 6 + (o + j)];
    JacobiEigen(3, B, ev, vec);
    e = std::sqrt(ev[2]);
    for (int i = 0; i < 3; ++i) dir[i] = vec[i * 3 + 2];
  };
  block(0, err.PositionError, err.PositionErrorDirection);
  block(3, err.OrientationError, err.OrientationErrorDirection);
  err.OrientationError = err.OrientationError / M_PI * 180.;
  return err;
}

}  // namespace orc
