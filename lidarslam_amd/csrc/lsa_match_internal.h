// lsa_match_internal.h -- what lsa_match.hip (targets, staged kernels, C ABI) and lsa_match_fused.hip (one launch
// per ICP iteration) share on the host side.
#pragma once
#include "lsa_knn.h"

namespace lsa
{

// one keypoint type's match, ready to be enqueued: parameters resolved, histogram block taken
struct MatchPrep
{
  int type;
  int ti;                       // target index (slot * 3 + type)
  const lsa_point_t* queries;
  int nq;
  MatchConst mc;
  float far_d2;                 // planes / blobs: the search may stop once the k-th neighbour is known to be farther
  int* hist;                    // device, 16 ints: [8] rejection histogram, [8], [9] hand-over counters
};

struct InterpConst;
// undistort: every keypoint is first moved by that motion interpolated at its own time, in place (lsa_undistort's step, folded
// into the search kernel); only when every keypoint of the set is among `preps` and is searched
// gate >= 0: the launch waits behind that gate (lsa_icp_gate) and takes pose -- and, gate_undistorts, the undistortion -- from it
int enqueue_fused_match(lsa_ctx* ctx, const MatchPrep* preps, int count, const double pose[16], hipStream_t st, const InterpConst* undistort = nullptr,
                        int gate = -1, bool gate_undistorts = false);

}  // namespace lsa
