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

// ---- the remaining result getters of Slam.h:141-189 ----
int lsa_slam_get_latency_compensated_world_transform(const lsa_slam* s, double T[16], double* time)
{
  if (!s || !T) return LSA_E_ARG;
  const lsa::host::Pose p = s->core.GetLatencyCompensatedWorldTransform(time);
  std::memcpy(T, p.m, sizeof(p.m));
  return LSA_OK;
}

int lsa_slam_set_base_to_lidar_offset(lsa_slam* s, const double T[16], int device_id)
{
  if (!s || !T) return LSA_E_ARG;
  lsa::host::Pose p;
  std::memcpy(p.m, T, sizeof(p.m));
  return s->core.SetBaseToLidarOffset(device_id, p);
}

int lsa_slam_get_base_to_lidar_offset(const lsa_slam* s, double T[16], int device_id)
{
  if (!s || !T || device_id < 0 || device_id > 255) return LSA_E_ARG;
  const lsa::host::Pose p = s->core.GetBaseToLidarOffset(device_id);
  std::memcpy(T, p.m, sizeof(p.m));
  return LSA_OK;
}

int lsa_slam_hint_next_stored_frame(lsa_slam* s, int slot)
{
  if (!s) return LSA_E_ARG;
  s->core.HintNextStoredFrame(slot);
  return LSA_OK;
}

int lsa_slam_clear_maps(lsa_slam* s)
{
  if (!s) return LSA_E_ARG;
  s->core.ClearMaps();
  return LSA_OK;
}

int lsa_slam_hint_next_frame(lsa_slam* s, const lsa_point_t* pts, int n)
{
  if (!s) return LSA_E_ARG;
  return s->core.HintNextFrame(pts, n);
}

int lsa_slam_set_extractor_param(lsa_slam* s, int device_id, const char* name, double value)
{
  if (!s || !name) return LSA_E_ARG;
  return s->core.SetExtractorParam(device_id, name, value);
}

int lsa_slam_get_extractor_param(const lsa_slam* s, int device_id, const char* name, double* value)
{
  if (!s || !name) return LSA_E_ARG;
  return s->core.GetExtractorParam(device_id, name, value);
}

int lsa_slam_add_frames(lsa_slam* s, const lsa_point_t* const* pts, const int* n, const uint64_t* stamp_us, const uint32_t* seq, int nframes)
{
  if (!s || !pts || !n || !stamp_us || nframes <= 0 || nframes > 16) return LSA_E_ARG;
  lsa::host::SlamCore::InputFrame frames[16];
  for (int i = 0; i < nframes; ++i) frames[i] = {pts[i], n[i], stamp_us[i], seq ? seq[i] : 0u};
  return s->core.AddFrames(frames, nframes);
}

int lsa_slam_set_world_transform_from_guess(lsa_slam* s, const double T[16])
{
  if (!s || !T) return LSA_E_ARG;
  lsa::host::Pose p;
  std::memcpy(p.m, T, sizeof(p.m));
  return s->core.SetWorldTransformFromGuess(p);
}

int lsa_slam_get_trajectory(const lsa_slam* s, double* poses, double* covariances, int capacity)
{
  if (!s || capacity < 0) return LSA_E_ARG;
  const auto& traj = s->core.LogTrajectory;
  const auto& covs = s->core.LogCovariances;
  const int n = static_cast<int>(traj.size());
  for (int i = 0; i < std::min(n, capacity); ++i)
  {
    if (poses)
    {
      std::memcpy(poses + 17 * i, traj[i].pose.m, 16 * sizeof(double));
      poses[17 * i + 16] = traj[i].time;
    }
    if (covariances)
    {
      // the covariances are logged only while LoggingTimeout != 0, newest last like the poses
      const long j = static_cast<long>(covs.size()) - n + i;
      if (j >= 0) std::memcpy(covariances + 36 * i, covs[j].data(), 36 * sizeof(double));
      else std::memset(covariances + 36 * i, 0, 36 * sizeof(double));
    }
  }
  return n;
}

int lsa_slam_get_debug_information(lsa_slam* s, double out[10])
{
  if (!s || !out) return LSA_E_ARG;
  return s->core.GetDebugInformation(out);
}

int lsa_slam_get_map(lsa_slam* s, int type, int clean, lsa_point_t* out, int capacity)
{
  if (!s || type < 0 || type > 2 || capacity < 0 || (capacity > 0 && !out)) return LSA_E_ARG;
  std::vector<lsa_point_t> pc;
  const int n = s->core.GetMap(type, clean != 0, pc);
  if (n < 0) return n;
  if (std::min(n, capacity) > 0) std::memcpy(out, pc.data(), static_cast<size_t>(std::min(n, capacity)) * sizeof(lsa_point_t));
  return n;
}

int lsa_slam_get_target_submap(lsa_slam* s, int type, lsa_point_t* out, int capacity)
{
  if (!s || type < 0 || type > 2 || capacity < 0 || (capacity > 0 && !out)) return LSA_E_ARG;
  std::vector<lsa_point_t> pc;
  const int n = s->core.GetTargetSubMap(type, pc);
  if (n < 0) return n;
  if (std::min(n, capacity) > 0) std::memcpy(out, pc.data(), static_cast<size_t>(std::min(n, capacity)) * sizeof(lsa_point_t));
  return n;
}

lsa_ctx* lsa_slam_context(lsa_slam* s) { return s ? s->core.Context() : nullptr; }

// ---- LidarSlam::RollingGrid on its own (include/lidarslam_amd.h, "the rolling voxel map") ----
struct lsa_rolling_grid
{
  lsa::host::RollingGrid grid;
};

lsa_rolling_grid* lsa_rolling_grid_create(void) { return new (std::nothrow) lsa_rolling_grid; }
void lsa_rolling_grid_destroy(lsa_rolling_grid* g) { delete g; }
int lsa_rolling_grid_set(lsa_rolling_grid* g, const char* name, double value)
{
  if (!g || !name) return LSA_E_ARG;
  const std::string n(name);
  if (n == "GridSize") g->grid.SetGridSize(static_cast<int>(value));
  else if (n == "VoxelResolution") g->grid.SetVoxelResolution(value);
  else if (n == "LeafSize") g->grid.SetLeafSize(value);
  else if (n == "MinFramesPerVoxel") g->grid.SetMinFramesPerVoxel(static_cast<unsigned int>(value));
  else if (n == "Sampling")
  {
    if (value < 0 || value > 4) return LSA_E_ARG;
    g->grid.SetSampling(static_cast<lsa::host::SamplingMode>(static_cast<int>(value)));
  }
  else if (n == "DecayingThreshold") g->grid.SetDecayingThreshold(value);
  else if (n == "AddThreads") g->grid.SetAddThreads(static_cast<int>(value));
  else if (n == "Ordered") g->grid.SetOrdered(value != 0);
  else return LSA_E_ARG;
  return LSA_OK;
}
void lsa_rolling_grid_reset(lsa_rolling_grid* g, const float position[3]) { if (g) g->grid.Reset(position); }
void lsa_rolling_grid_clear(lsa_rolling_grid* g) { if (g) g->grid.Clear(); }
int lsa_rolling_grid_size(const lsa_rolling_grid* g) { return g ? static_cast<int>(g->grid.Size()) : LSA_E_ARG; }
void lsa_rolling_grid_roll(lsa_rolling_grid* g, const float min_point[3], const float max_point[3])
{
  if (g && min_point && max_point) g->grid.Roll(min_point, max_point);
}
int lsa_rolling_grid_add(lsa_rolling_grid* g, const lsa_point_t* pts, int n, int fixed, double current_time, int roll)
{
  if (!g || n < 0 || (n > 0 && !pts)) return LSA_E_ARG;
  g->grid.Add(pts, static_cast<std::size_t>(n), fixed != 0, current_time, roll != 0);
  return LSA_OK;
}
void lsa_rolling_grid_clear_old_points(lsa_rolling_grid* g, double current_time) { if (g) g->grid.ClearOldPoints(current_time); }
int lsa_rolling_grid_get(const lsa_rolling_grid* g, int clean, lsa_point_t* out, int capacity)
{
  if (!g || capacity < 0 || (capacity > 0 && !out)) return LSA_E_ARG;
  const lsa::host::RollingGrid::PointCloud pc = g->grid.Get(clean != 0);
  const int n = std::min<int>(capacity, static_cast<int>(pc.size()));
  if (n > 0) std::memcpy(out, pc.data(), static_cast<std::size_t>(n) * sizeof(lsa_point_t));
  return n;
}
int lsa_rolling_grid_build_submap(lsa_rolling_grid* g, const float min_point[3], const float max_point[3], int min_nb_points)
{
  if (!g || ((min_point == nullptr) != (max_point == nullptr))) return LSA_E_ARG;
  if (min_point) g->grid.BuildSubMap(min_point, max_point, min_nb_points);
  else g->grid.BuildSubMap();
  return static_cast<int>(g->grid.SubMapSize());
}
int lsa_rolling_grid_submap_valid(const lsa_rolling_grid* g) { return g && g->grid.IsSubMapValid() ? 1 : 0; }
int lsa_rolling_grid_submap(const lsa_rolling_grid* g, lsa_point_t* out, int capacity)
{
  if (!g || capacity < 0 || (capacity > 0 && !out)) return LSA_E_ARG;
  const int n = std::min<int>(capacity, static_cast<int>(g->grid.SubMapSize()));
  if (n > 0) std::memcpy(out, g->grid.SubMapData(), static_cast<std::size_t>(n) * sizeof(lsa_point_t));
  return n;
}

}  // extern "C"
