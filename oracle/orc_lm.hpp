// ORACLE -- TEST INFRASTRUCTURE ONLY (see orc_math.hpp header).  PARITY UNPINNED.
//
// CPU restatement of LidarSlam::LocalOptimizer (slam_lib/src/LocalOptimizer.cxx)
// and of the part of Ceres (unpinned `master` in the reference CI image, >= 2.0)
// that it drives: trust-region Levenberg-Marquardt minimiser with Jacobi
// scaling, TukeyLoss/ScaledLoss robustification through the Triggs corrector,
// and the DENSE_SVD covariance.  Ceres is not under /root/reference; its
// published algorithm is restated here (recalled from upstream -- not
// verifiable offline).
#pragma once
#include <vector>
#include "orc_math.hpp"
#include "orc_matcher.hpp"

namespace orc
{

struct LMSummary
{
  int num_successful_steps = 0;    // counts iteration 0, like ceres::Solver::Summary
  int num_unsuccessful_steps = 0;
  int num_iterations = 0;
  int num_evaluations = 0;         // cost(+jacobian) evaluations, for the bytes model of SURVEY 8d
  double initial_cost = 0., final_cost = 0.;
  const char* message = "";
};

// LocalOptimizer::RegistrationError (LocalOptimizer.h:36-52)
struct RegistrationError
{
  double PositionError = 0.;
  double PositionErrorDirection[3] = {0, 0, 0};
  double OrientationError = 0.;
  double OrientationErrorDirection[3] = {0, 0, 0};
  double Covariance[36] = {0};
};

// value of the robustified problem at w: cost = 1/2 sum rho, g = J^T r, H = J^T J
// (corrected by sqrt(rho')), all in the reference's residual-block order
struct NormalEq
{
  double cost = 0.;
  double g[6] = {0};
  double H[36] = {0};
  int nValid = 0;
};

// Evaluate residual blocks [begin,end) at w (X,Y,Z,rX,rY,rZ); adds into out.
void EvaluateResiduals(const Residual* res, size_t n, const double w[6], bool jacobians, NormalEq& out, int nbThreads = 1);

class LocalOptimizer
{
public:
  void SetTwoDMode(bool b) { TwoDMode = b; }
  void SetLMMaxIter(unsigned n) { LMMaxIter = n; }
  void SetNbThreads(unsigned n) { NbThreads = n; }
  void SetPosePrior(const Iso& prior) { iso_to_xyzrpy(prior, PoseArray); }   // LocalOptimizer.cxx:44-48
  void AddResiduals(const std::vector<Residual>& r) { for (const auto& x : r) if (x.valid) Residuals.push_back(x); }  // :59-62 + :82-86
  LMSummary Solve();                                                          // :74-102
  Iso GetOptimizedPose() const { return xyzrpy_to_iso(PoseArray); }           // :105-109
  RegistrationError EstimateRegistrationError();                              // :112-140
  const double* GetPoseArray() const { return PoseArray; }
  size_t NbResiduals() const { return Residuals.size(); }

private:
  bool TwoDMode = false;
  unsigned NbThreads = 1;
  unsigned LMMaxIter = 15;
  double PoseArray[6] = {0, 0, 0, 0, 0, 0};
  std::vector<Residual> Residuals;
};

// 6x6 helpers shared with tests
bool CholeskySolve(int n, const double* A, const double* b, double* x);
void JacobiEigen(int n, const double* A, double* evals, double* evecs);  // ascending, columns = vectors

}  // namespace orc
