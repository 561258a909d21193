// lsa_slam_capi.cpp -- pipeline-level C ABI (lsa_slam_*): LidarSlam::Slam behind a handle.
// Mirrors Slam::AddFrame / GetWorldTransform / GetKeypoints / GetRegisteredFrame
// (slam_lib/include/LidarSlam/Slam.h:111-178).
#include <algorithm>
#include <cstring>
#include "lsa_slam_core.h"

using lsa::host::SlamCore;

struct lsa_slam
{
  explicit lsa_slam(int device) : core(device) {}
  SlamCore core;
  std::vector<lsa_point_t> scratch;
};

extern "C" {

int lsa_slam_create(int device_id, lsa_slam** out)
{
  if (!out) return LSA_E_ARG;
  *out = nullptr;
  lsa_slam* s = new lsa_slam(device_id);
  if (!s->core.Ok())
  {
    delete s;
    return LSA_E_NO_DEVICE;  // loud failure: the device path is the only path
  }
  *out = s;
  return LSA_OK;
}

void lsa_slam_destroy(lsa_slam* s) { delete s; }

const char* lsa_slam_last_error(const lsa_slam* s) { return s ? s->core.Error().c_str() : "null handle"; }

int lsa_slam_set_param(lsa_slam* s, const char* name, double value)
{
  if (!s || !name) return LSA_E_ARG;
  return s->core.SetParam(name, value);
}

int lsa_slam_get_param(const lsa_slam* s, const char* name, double* value)
{
  if (!s || !name) return LSA_E_ARG;
  return s->core.GetParam(name, value);
}

void lsa_slam_reset(lsa_slam* s, int reset_log)
{
  if (s) s->core.Reset(reset_log != 0);
}

int lsa_slam_add_frame(lsa_slam* s, const lsa_point_t* pts, int n, uint64_t stamp_us, uint32_t seq)
{
  if (!s) return LSA_E_ARG;
  return s->core.AddFrame(pts, n, stamp_us, seq);
}

int lsa_slam_store_frame(lsa_slam* s, int slot, const lsa_point_t* pts, int n)
{
  if (!s) return LSA_E_ARG;
  return lsa_frame_store_put(s->core.Context(), slot, pts, n);
}

int lsa_slam_add_stored_frame(lsa_slam* s, int slot, uint64_t stamp_us, uint32_t seq)
{
  if (!s) return LSA_E_ARG;
  return s->core.AddStoredFrame(slot, stamp_us, seq);
}

int lsa_slam_get_world_transform(const lsa_slam* s, double T[16], double* time)
{
  if (!s || !T) return LSA_E_ARG;
  const lsa::host::Pose p = s->core.GetWorldTransform(time);
  std::memcpy(T, p.m, sizeof(p.m));
  return LSA_OK;
}

int lsa_slam_get_covariance(const lsa_slam* s, double cov[36])
{
  if (!s || !cov) return LSA_E_ARG;
  std::memcpy(cov, s->core.GetTransformCovariance().data(), 36 * sizeof(double));
  return LSA_OK;
}

int lsa_slam_get_keypoints(lsa_slam* s, int type, int which, lsa_point_t* out, int capacity)
{
  if (!s || type < 0 || type > 2 || (!out && capacity > 0)) return LSA_E_ARG;
  int n = which == 2 ? s->core.GetRawKeypoints(type, s->scratch) : s->core.GetKeypoints(type, which == 1, s->scratch);
  if (n < 0) return n;
  n = std::min(n, capacity);
  if (n > 0) std::memcpy(out, s->scratch.data(), (size_t)n * sizeof(lsa_point_t));
  return n;
}

int lsa_slam_get_registered_frame(lsa_slam* s, lsa_point_t* out, int capacity)
{
  if (!s || (!out && capacity > 0)) return LSA_E_ARG;
  int n = s->core.GetRegisteredFrame(s->scratch);
  if (n < 0) return n;
  n = std::min(n, capacity);
  if (n > 0) std::memcpy(out, s->scratch.data(), (size_t)n * sizeof(lsa_point_t));
  return n;
}

int lsa_slam_get_match_status(lsa_slam* s, int localization, int type, uint8_t* status, double* weights, int capacity)
{
  if (!s || type < 0 || type > 2) return LSA_E_ARG;
  const lsa::host::MatchDebug& d = s->core.GetMatchDebug(localization != 0, type);
  const int n = std::min<int>(capacity, d.status.size());
  if (n > 0 && status) std::memcpy(status, d.status.data(), n);
  if (n > 0 && weights) std::memcpy(weights, d.weights.data(), (size_t)n * sizeof(double));
  return n;
}

int lsa_slam_get_stats(const lsa_slam* s, double out[16])
{
  if (!s || !out) return LSA_E_ARG;
  const lsa::host::FrameStats& t = s->core.Stats;
  const double v[16] = {t.total, t.extract, t.ego_icp, t.ego_lm, t.loc_icp, t.loc_lm, t.undistort, t.submap, t.maps,
                        (double)t.ego_iters, (double)t.loc_iters, (double)t.lm_evals, (double)s->core.TotalMatchedKeypoints,
                        (double)s->core.KfCounter, t.maps_wait, t.maps_async};
  std::memcpy(out, v, sizeof(v));
  return LSA_OK;
}

lsa_ctx* lsa_slam_context(lsa_slam* s) { return s ? s->core.Context() : nullptr; }

}  // extern "C"
