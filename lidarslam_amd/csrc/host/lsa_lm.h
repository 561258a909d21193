// lsa_lm.h -- LidarSlam::LocalOptimizer for the MI355X path
// (slam_lib/include/LidarSlam/LocalOptimizer.h, slam_lib/src/LocalOptimizer.cxx).
//
// The residual blocks live on the device (lsa_match wrote them); every evaluation the
// trust-region loop asks for is one lsa_accumulate call (cost, gradient, J^T J reduced on
// the GPU, 29 doubles read back).  The 6-dof Levenberg-Marquardt control flow of Ceres
// (unpinned master >= 2.0 in the reference CI) is restated on the host: Jacobi scaling,
// LM diagonal, step acceptance, radius update, termination tests.
#pragma once
#include <array>
#include "../lsa_ctx.h"
#include "lsa_hostmath.h"

namespace lsa
{
namespace host
{

// counterpart of ceres::Solver::Summary fields the reference reads (Slam.cxx:950, 1151)
struct SolveSummary
{
  int num_successful_steps = 0;
  int num_unsuccessful_steps = 0;
  int num_iterations = 0;
  int num_evaluations = 0;
  double initial_cost = 0., final_cost = 0.;
  int num_matches = 0;      // residual blocks (successful matches) the problem was built from
  bool skipped = false;     // fewer than SetMinMatches(): nothing was optimised, the prior is returned
  const char* message = "";
};

// LocalOptimizer::RegistrationError (LocalOptimizer.h:36-52)
struct RegistrationError
{
  double PositionError = 0.;
  double PositionErrorDirection[3] = {0, 0, 0};
  double OrientationError = 0.;
  double OrientationErrorDirection[3] = {0, 0, 0};
  std::array<double, 36> Covariance{};
};

class LocalOptimizer
{
public:
  explicit LocalOptimizer(lsa_ctx* ctx) : Ctx(ctx) {}
  void SetTwoDMode(bool b) { TwoDMode = b; }
  void SetLMMaxIter(unsigned n) { LMMaxIter = n; }
  void SetPosePrior(const Pose& prior) { ToXYZRPY(prior, PoseArray); HaveFinal = false; }
  // residual blocks = the device records of the last lsa_match of every type in the mask
  void UseDeviceResiduals(unsigned typeMask) { TypeMask = typeMask; }
  // Slam.cxx:919-923 / 1098-1107 skip the optimisation when too few keypoints matched.  The count comes
  // back with the first evaluation, so the caller does not have to read the match histograms first.
  void SetMinMatches(unsigned n) { MinMatches = n; }
  // the trust-region loop on the device, one launch per solve (lsa_solve_device); when off, or when the device gives
  // up, the loop runs here with one launch per evaluation.  Same decisions from the same sums either way.
  void SetDeviceLoop(bool b) { DeviceLoop = b; }
  int Solve(SolveSummary& summary);
  // Solve in two halves (device loop only): Begin enqueues the solve -- gated: behind the gate enqueued last (lsa_icp_gate),
  // which will hand over the start point; SetPosePrior must then have been called with that very pose before End --, End
  // waits for the result of the oldest solve begun on this context (by this object or an earlier one with the same
  // settings).  End returns LSA_E_GATE when the gate gave up (nothing ran); a device solve that gave up itself is redone
  // on the host -- End then returns 1: the iterations enqueued ahead have been called off.
  int Begin(bool gated);
  int End(SolveSummary& summary);
  Pose GetOptimizedPose() const { return FromXYZRPY(PoseArray); }
  int EstimateRegistrationError(RegistrationError& err);

private:
  lsa_ctx* Ctx;
  unsigned TypeMask = 7;
  bool TwoDMode = false;
  unsigned LMMaxIter = 15;
  unsigned MinMatches = 0;
  double PoseArray[6] = {0, 0, 0, 0, 0, 0};
  bool DeviceLoop = true;
  // normal equations at PoseArray as the device solve returned them (EstimateRegistrationError needs no evaluation then)
  bool HaveFinal = false;
  double FinalH[36];
  int SolveOnHost(SolveSummary& summary);
  void TakeResult(const lsa_solve_result_t& r, SolveSummary& summary);
};

}  // namespace host
}  // namespace lsa
