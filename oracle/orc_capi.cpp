// ORACLE -- TEST INFRASTRUCTURE ONLY (see orc_math.hpp header).  PARITY UNPINNED.
//
// extern "C" surface of the oracle so that tests/ (ctypes) and bench.py's
// cpu_baseline leg can drive it.  The POD parameter structs are the ones of
// include/lidarslam_amd.h so that a test passes the very same bytes to the HIP
// path and to the oracle.
#include <cstring>
#include <map>
#include <string>
#include <vector>
#include "../include/lidarslam_amd.h"
#include "orc_slam.hpp"

using namespace orc;
static_assert(sizeof(lsa_point_t) == sizeof(Point), "point layout");

namespace
{
Iso IsoFromRowMajor(const double T[16])
{
  Iso r;
  for (int i = 0; i < 3; ++i)
  {
    for (int j = 0; j < 3; ++j) r.R[i * 3 + j] = T[i * 4 + j];
    r.t[i] = T[i * 4 + 3];
  }
  return r;
}
void IsoToRowMajor(const Iso& a, double T[16])
{
  for (int i = 0; i < 3; ++i)
  {
    for (int j = 0; j < 3; ++j) T[i * 4 + j] = a.R[i * 3 + j];
    T[i * 4 + 3] = a.t[i];
  }
  T[12] = T[13] = T[14] = 0; T[15] = 1;
}
void ApplyExtractParams(Extractor& e, const lsa_extract_params_t* p)
{
  e.P.NeighborWidth = p->neighbor_width;
  e.P.MinDistanceToSensor = p->min_distance_to_sensor;
  e.P.MinBeamSurfaceAngle = p->min_beam_surface_angle;
  e.P.PlaneSinAngleThreshold = p->plane_sin_angle_threshold;
  e.P.EdgeSinAngleThreshold = p->edge_sin_angle_threshold;
  e.P.DistToLineThreshold = p->dist_to_line_threshold;
  e.P.EdgeDepthGapThreshold = p->edge_depth_gap_threshold;
  e.P.EdgeSaliencyThreshold = p->edge_saliency_threshold;
  e.P.EdgeIntensityGapThreshold = p->edge_intensity_gap_threshold;
}
MatchParams ToMatchParams(const lsa_match_params_t* p, int nbThreads)
{
  MatchParams m;
  m.NbThreads = nbThreads;
  m.SingleEdgePerRing = p->single_edge_per_ring != 0;
  m.MaxNeighborsDistance = p->max_neighbors_distance;
  m.EdgeNbNeighbors = p->edge_nb_neighbors;
  m.EdgeMinNbNeighbors = p->edge_min_nb_neighbors;
  m.EdgeMaxModelError = p->edge_max_model_error;
  m.PlaneNbNeighbors = p->plane_nb_neighbors;
  m.PlanarityThreshold = p->planarity_threshold;
  m.PlaneMaxModelError = p->plane_max_model_error;
  m.BlobNbNeighbors = p->blob_nb_neighbors;
  m.SaturationDistance = p->saturation_distance;
  return m;
}
struct ExtractorHandle
{
  Extractor e;
  std::vector<Point> scan;
};
struct SlamHandle
{
  Slam s;
  std::vector<Point> frame;
  std::vector<std::vector<Point>> frames;  // AddFrames
};
std::vector<Residual> RecordsToResiduals(const double* records, const uint8_t* status, int n, double sat)
{
  std::vector<Residual> out(n);
  for (int i = 0; i < n; ++i)
  {
    if (status[i] != SUCCESS) continue;
    Residual& r = out[i];
    r.valid = true;
    std::memcpy(r.A, records + 16 * i, 9 * sizeof(double));
    std::memcpy(r.P, records + 16 * i + 9, 3 * sizeof(double));
    std::memcpy(r.X, records + 16 * i + 12, 3 * sizeof(double));
    r.weight = records[16 * i + 15];
    r.sat = sat;
  }
  return out;
}
}  // namespace

extern "C" {

// ---- extractor ------------------------------------------------------------
void* orc_extractor_create() { return new ExtractorHandle; }
void orc_extractor_destroy(void* h) { delete (ExtractorHandle*)h; }
void orc_extractor_set_threads(void* h, int n) { ((ExtractorHandle*)h)->e.P.NbThreads = n; }
float orc_extractor_get_azimuthal_resolution(void* h) { return ((ExtractorHandle*)h)->e.AzimuthalResolution; }
void orc_extractor_set_azimuthal_resolution(void* h, float v) { ((ExtractorHandle*)h)->e.AzimuthalResolution = v; }
int orc_extractor_compute(void* h, const lsa_extract_params_t* p, const lsa_point_t* pts, int n, int counts[3])
{
  ExtractorHandle* eh = (ExtractorHandle*)h;
  ApplyExtractParams(eh->e, p);
  eh->scan.assign((const Point*)pts, (const Point*)pts + n);
  eh->e.ComputeKeyPoints(eh->scan);
  for (int k = 0; k < 3; ++k) counts[k] = (int)eh->e.Keypoints[k].size();
  return 0;
}
int orc_extractor_keypoints(void* h, int type, lsa_point_t* out, int capacity)
{
  const auto& k = ((ExtractorHandle*)h)->e.Keypoints[type];
  int n = std::min<int>(capacity, k.size());
  std::memcpy(out, k.data(), n * sizeof(Point));
  return n;
}
int orc_extractor_debug(void* h, int id, float* out, int capacity)
{
  std::vector<float> v = ((ExtractorHandle*)h)->e.DebugArray(id);
  int n = std::min<int>(capacity, v.size());
  std::memcpy(out, v.data(), n * sizeof(float));
  return n;
}
int orc_extractor_nb_rings(void* h) { return ((ExtractorHandle*)h)->e.NbLaserRings; }

// ---- kNN --------------------------------------------------------------------
int orc_knn(const lsa_point_t* tgt, int m, const double* queries, int nq, int k, int* idx, float* d2, int* counts)
{
  std::vector<Point> cloud((const Point*)tgt, (const Point*)tgt + m);
  KDTree tree;
  tree.Reset(&cloud);
  for (int i = 0; i < nq; ++i)
    counts[i] = tree.KnnSearch(queries + 3 * i, k, idx + (size_t)i * k, d2 + (size_t)i * k);
  return 0;
}
// brute force with the same (distance, index) order, to check the tree itself
int orc_knn_brute(const lsa_point_t* tgt, int m, const double* queries, int nq, int k, int* idx, float* d2, int* counts)
{
  const Point* c = (const Point*)tgt;
  std::vector<std::pair<float, int>> all(m);
  for (int i = 0; i < nq; ++i)
  {
    float q[3] = {(float)queries[3 * i], (float)queries[3 * i + 1], (float)queries[3 * i + 2]};
    for (int j = 0; j < m; ++j) all[j] = {KDTree::Dist2(q, c[j]), j};
    int kk = std::min(k, m);
    std::partial_sort(all.begin(), all.begin() + kk, all.end());
    for (int j = 0; j < kk; ++j) { idx[(size_t)i * k + j] = all[j].second; d2[(size_t)i * k + j] = all[j].first; }
    counts[i] = kk;
  }
  return 0;
}

// ---- matcher ----------------------------------------------------------------
int orc_match(const lsa_point_t* cur, int n, const lsa_point_t* tgt, int m, int type, const lsa_match_params_t* mp,
              const double pose[16], int nbThreads, uint8_t* status, double* weights, double* records, int hist[8])
{
  std::vector<Point> c((const Point*)cur, (const Point*)cur + n), t((const Point*)tgt, (const Point*)tgt + m);
  KDTree tree;
  tree.Reset(&t);
  KeypointsMatcher matcher(ToMatchParams(mp, nbThreads), IsoFromRowMajor(pose));
  MatchingResults r = matcher.BuildMatchResiduals(c, tree, (Keypoint)type);
  for (int i = 0; i < n; ++i)
  {
    status[i] = r.Rejections[i];
    weights[i] = r.Weights[i];
    if (records)
    {
      double* o = records + 16 * (size_t)i;
      std::memset(o, 0, 16 * sizeof(double));
      if (r.Residuals[i].valid)
      {
        std::memcpy(o, r.Residuals[i].A, 9 * sizeof(double));
        std::memcpy(o + 9, r.Residuals[i].P, 3 * sizeof(double));
        std::memcpy(o + 12, r.Residuals[i].X, 3 * sizeof(double));
        o[15] = r.Residuals[i].weight;
      }
    }
  }
  for (int s = 0; s < 8; ++s) hist[s] = r.RejectionsHistogram[s];
  return 0;
}

// ---- normal equations / LM -----------------------------------------------------
int orc_accumulate(const double* records, const uint8_t* status, int n, double sat, const double w[6], int jac, double* cost,
                   double g[6], double H[36], int* nvalid)
{
  std::vector<Residual> res = RecordsToResiduals(records, status, n, sat);
  NormalEq e;
  EvaluateResiduals(res.data(), res.size(), w, jac != 0, e);
  *cost = e.cost;
  std::memcpy(g, e.g, sizeof(e.g));
  std::memcpy(H, e.H, sizeof(e.H));
  *nvalid = e.nValid;
  return 0;
}
// summary[0] num_successful_steps, [1] num_unsuccessful, [2] iterations, [3] evaluations
int orc_lm_solve(const double* records, const uint8_t* status, int n, double sat, const double pose_in[16], int maxIter,
                 int twoD, double pose_out[16], double w_out[6], int summary[4], double costs[2])
{
  LocalOptimizer opt;
  opt.SetTwoDMode(twoD != 0);
  opt.SetLMMaxIter(maxIter);
  opt.SetPosePrior(IsoFromRowMajor(pose_in));
  opt.AddResiduals(RecordsToResiduals(records, status, n, sat));
  LMSummary s = opt.Solve();
  IsoToRowMajor(opt.GetOptimizedPose(), pose_out);
  std::memcpy(w_out, opt.GetPoseArray(), 6 * sizeof(double));
  summary[0] = s.num_successful_steps; summary[1] = s.num_unsuccessful_steps;
  summary[2] = s.num_iterations; summary[3] = s.num_evaluations;
  costs[0] = s.initial_cost; costs[1] = s.final_cost;
  return 0;
}
int orc_covariance(const double* records, const uint8_t* status, int n, double sat, const double pose[16], double cov[36], double err[2])
{
  LocalOptimizer opt;
  opt.SetPosePrior(IsoFromRowMajor(pose));
  opt.AddResiduals(RecordsToResiduals(records, status, n, sat));
  RegistrationError e = opt.EstimateRegistrationError();
  std::memcpy(cov, e.Covariance, sizeof(e.Covariance));
  err[0] = e.PositionError; err[1] = e.OrientationError;
  return 0;
}

// ---- elementary functions (same ids as lsa_selftest_math) ----
// 1: the eigen-solver and the slerp call libm (glibc) like the reference's PCL / Eigen do, 0: lsa_pmath.h (default)
void orc_set_libm_trig(int on) { orc::libm_trig() = on ? 1 : 0; }

int orc_math(int fn, const double* x, const double* y, int n, double* out)
{
  for (int i = 0; i < n; ++i)
  {
    switch (fn)
    {
      case 0: out[i] = lsa_sin(x[i]); break;
      case 1: out[i] = lsa_cos(x[i]); break;
      case 2: out[i] = lsa_atan2(y[i], x[i]); break;
      case 3: out[i] = (double)std::sqrt((float)x[i]); break;
      case 4: out[i] = (double)((float)x[i] / (float)y[i]); break;
      case 5: out[i] = std::sqrt(x[i]); break;
      case 6: out[i] = x[i] / y[i]; break;
      case 7: out[i] = lsa_asin(x[i]); break;
      case 8: out[i] = lsa_acos(x[i]); break;
      default: return -1;
    }
  }
  return 0;
}

// ---- undistortion ------------------------------------------------------------------
int orc_undistort(lsa_point_t* pts, int n, const double H0[16], const double H1[16], double t0, double t1)
{
  Interpolator it;
  it.SetTimes(t0, t1);
  it.SetTransforms(IsoFromRowMajor(H0), IsoFromRowMajor(H1));
  Point* p = (Point*)pts;
  for (int i = 0; i < n; ++i) transform_point(p[i], it(p[i].time));
  return 0;
}
// ---- the pose algebra between two ICP iterations ------------------------------------------------------------------
// What Slam::ComputeEgoMotion / Localization do with a solve's parameters before the next iteration (Slam.cxx:940-950,
// 1134-1151): Utils::XYZRPYtoIsometry, the next start point (LocalOptimizer::SetPosePrior: IsometryToXYZRPY of that pose),
// and -- refine -- Slam::RefineUndistortion (Slam.cxx:1271-1285, 1322-1352) with its LinearTransformInterpolator, by THIS
// restatement's own functions (xyzrpy_to_iso, iso_to_xyzrpy, linear_interpolation, Interpolator, slerp_prepare).  Laid out
// as the block the product's solve leaves on the device for the next iteration (words: go | pose R[9] t[3] | start point
// [6] | qa[4] qb[4] d theta sin_theta trans0[3] trans1[3] time0 time1 | h0 R[9] t[3] | {linear, invalid}), so that a test
// can hold the product's arithmetic (lsa_posemath.h, host and device) against this one word for word.
// motion: Time0, Time1, Rot0 (w x y z), Rot1, Trans0[3], Trans1[3] of WithinFrameMotion before; motion_after: afterwards.
int orc_icp_link(const double x[6], int refine, int have_log, double prev_time, double cur_time, double max_ratio, const double previous_world[16],
                 const double motion[16], unsigned long long words[64], double motion_after[16])
{
  std::memset(words, 0, 64 * sizeof(unsigned long long));
  std::memcpy(motion_after, motion, 16 * sizeof(double));
  const Iso Tworld = xyzrpy_to_iso(x);
  double x0[6];
  iso_to_xyzrpy(Tworld, x0);
  double* w = reinterpret_cast<double*>(words);
  words[0] = 1ull;
  std::memcpy(w + 1, Tworld.R, sizeof(Tworld.R));
  std::memcpy(w + 10, Tworld.t, sizeof(Tworld.t));
  std::memcpy(w + 13, x0, sizeof(x0));
  if (!refine) return 0;
  Interpolator within;
  within.Time0 = motion[0]; within.Time1 = motion[1];
  within.Rot0 = {motion[2], motion[3], motion[4], motion[5]};
  within.Rot1 = {motion[6], motion[7], motion[8], motion[9]};
  std::memcpy(within.Trans0, motion + 10, sizeof(within.Trans0));
  std::memcpy(within.Trans1, motion + 13, sizeof(within.Trans1));
  const Iso PreviousTworld = IsoFromRowMajor(previous_world);
  auto InterpolateScanPose = [&](double time) -> Iso {  // Slam.cxx:1271-1285
    if (!have_log) return Tworld;
    if (std::abs(time / (cur_time - prev_time)) > max_ratio) return Tworld;
    return linear_interpolation(PreviousTworld, Tworld, cur_time + time, prev_time, cur_time);
  };
  // Slam.cxx:1322-1352
  const Iso previousBaseBegin = within.GetH0();
  const Iso previousBaseEnd = within.GetH1();
  const Iso worldToBaseBegin = InterpolateScanPose(within.Time0);
  const Iso worldToBaseEnd = InterpolateScanPose(within.Time1);
  const Iso baseToWorld = iso_inverse(Tworld);
  const Iso newBaseBegin = iso_mul(baseToWorld, worldToBaseBegin);
  const Iso newBaseEnd = iso_mul(baseToWorld, worldToBaseEnd);
  within.SetTransforms(newBaseBegin, newBaseEnd);
  Interpolator interp = within;
  interp.SetTransforms(iso_mul(newBaseBegin, iso_inverse(previousBaseBegin)), iso_mul(newBaseEnd, iso_inverse(previousBaseEnd)));
  const SlerpConst sc = slerp_prepare(interp.Rot0, interp.Rot1);
  const Iso h0 = interp.GetH0();
  double* c = w + 19;
  const double qa[4] = {interp.Rot0.w, interp.Rot0.x, interp.Rot0.y, interp.Rot0.z}, qb[4] = {interp.Rot1.w, interp.Rot1.x, interp.Rot1.y, interp.Rot1.z};
  std::memcpy(c, qa, sizeof(qa)); std::memcpy(c + 4, qb, sizeof(qb));
  c[8] = sc.d; c[9] = sc.theta; c[10] = sc.sin_theta;
  std::memcpy(c + 11, interp.Trans0, 3 * sizeof(double)); std::memcpy(c + 14, interp.Trans1, 3 * sizeof(double));
  c[17] = interp.Time0; c[18] = interp.Time1;
  std::memcpy(c + 19, h0.R, sizeof(h0.R)); std::memcpy(c + 28, h0.t, sizeof(h0.t));
  const int flags[2] = {sc.linear ? 1 : 0, interp.IsInvalid ? 1 : 0};
  std::memcpy(c + 31, flags, sizeof(flags));
  double* o = motion_after;
  o[0] = within.Time0; o[1] = within.Time1;
  o[2] = within.Rot0.w; o[3] = within.Rot0.x; o[4] = within.Rot0.y; o[5] = within.Rot0.z;
  o[6] = within.Rot1.w; o[7] = within.Rot1.x; o[8] = within.Rot1.y; o[9] = within.Rot1.z;
  std::memcpy(o + 10, within.Trans0, 3 * sizeof(double)); std::memcpy(o + 13, within.Trans1, 3 * sizeof(double));
  return 0;
}

int orc_transform(lsa_point_t* pts, int n, const double T[16])
{
  Iso a = IsoFromRowMajor(T);
  Point* p = (Point*)pts;
  for (int i = 0; i < n; ++i) transform_point(p[i], a);
  return 0;
}

// Confidence::LCPEstimator (slam_lib/src/ConfidenceEstimators.cxx:27-65) on an already registered cloud:
// every (1 / ratio)-th point, 1-NN into each non-empty target, best Gaussian score with sigma = leaf / 3,
// mean over the sampled points.  The reference sums in float under an OpenMP reduction (order not defined);
// here sequentially.  Returns -1 like the reference when nothing can be estimated.
float orc_lcp(const lsa_point_t* cloud, int n, float ratio, const lsa_point_t* const tgt[3], const int m[3], const double leaf[3])
{
  const int nbPoints = (int)(n * ratio);  // size_t * float -> float -> int (:33)
  bool any = false;
  KDTree tree[3];
  std::vector<Point> clouds[3];
  for (int k = 0; k < 3; ++k)
    if (tgt[k] && m[k] > 0)
    {
      clouds[k].assign((const Point*)tgt[k], (const Point*)tgt[k] + m[k]);
      tree[k].Reset(&clouds[k]);
      any = true;
    }
  if (nbPoints == 0 || !any) return -1.f;
  const Point* c = (const Point*)cloud;
  float lcp = 0.f;
  for (int i = 0; i < nbPoints; ++i)
  {
    const Point& point = c[(size_t)(i / ratio)];  // cloud->at(n / subsamplingRatio) (:43)
    float bestProba = 0.f;
    for (int k = 0; k < 3; ++k)
    {
      if (clouds[k].empty()) continue;
      int nnIndex[2];
      float nnSqDist[2];
      const float q[3] = {point.x, point.y, point.z};
      if (tree[k].KnnSearch(q, 1, nnIndex, nnSqDist))
      {
        const float sqLCPThreshold = (float)std::pow(leaf[k] / 3.f, 2);                 // (:55)
        const float currentProba = std::exp(-nnSqDist[0] / (2.f * sqLCPThreshold));     // float exp (:56)
        if (currentProba > bestProba) bestProba = currentProba;
      }
    }
    lcp += bestProba;
  }
  return lcp / nbPoints;
}

// VelodyneToLidarNode::Callback (ros_wrapping/lidar_conversions/src/VelodyneToLidarNode.cxx:52-112) with
// Utils::SpinningFrameAdvancementEstimator (ros_wrapping/lidar_conversions/src/Utilities.h:62-114): driver
// records (float x, y, z, intensity, time; uint16 ring at the given byte offsets) -> LidarPoint.
// layout = {point_step, off_x, off_y, off_z, off_intensity, off_ring, off_time}.  Returns 1 when the time
// field was usable, 0 when the time was built from the azimuth advancement, -1 for an empty cloud.
int orc_velodyne_to_lidar(const void* records, int n, const int32_t layout[7], const uint16_t* mapping, int mapping_len, int device_id, double rpm,
                          int timestamp_first_packet, lsa_point_t* out)
{
  if (n <= 0) return -1;  // "Input Velodyne pointcloud is empty : frame ignored." (:56-60)
  const unsigned char* raw = (const unsigned char*)records;
  const int step = layout[0];
  auto f32 = [&](int i, int off) { float v; std::memcpy(&v, raw + (size_t)i * step + off, 4); return v; };
  const bool isTimeValid = f32(n - 1, layout[6]) - f32(0, layout[6]) > 1e-8;  // (:74)
  double initAdvancement = 0.;
  std::map<int, double> previousAdvancementPerRing;
  for (int i = 0; i < n; ++i)
  {
    lsa_point_t p;
    std::memset(&p, 0, sizeof(p));
    uint16_t ring;
    std::memcpy(&ring, raw + (size_t)i * step + layout[5], 2);
    p.x = f32(i, layout[1]); p.y = f32(i, layout[2]); p.z = f32(i, layout[3]); p.w = 1.f;
    p.intensity = f32(i, layout[4]);
    p.laser_id = mapping_len > 0 ? mapping[ring] : ring;  // (:90)
    p.device_id = (uint8_t)device_id;
    if (isTimeValid) p.time = f32(i, layout[6]);  // (:95-96)
    else
    {
      // SpinningFrameAdvancementEstimator::operator() (Utilities.h:88-108)
      double pointAdvancement = (M_PI - std::atan2(p.y, p.x)) / (2 * M_PI);
      if (previousAdvancementPerRing.empty()) initAdvancement = pointAdvancement;
      auto wrapMax = [](double x, double max) { return std::fmod(max + std::fmod(x, max), max); };
      double frameAdvancement = wrapMax(pointAdvancement - initAdvancement, 1.);
      if (frameAdvancement < previousAdvancementPerRing[p.laser_id]) frameAdvancement += 1.;
      previousAdvancementPerRing[p.laser_id] = frameAdvancement;
      p.time = (timestamp_first_packet ? frameAdvancement : frameAdvancement - 1) / rpm * 60.;  // (:106)
    }
    out[i] = p;
  }
  return isTimeValid ? 1 : 0;
}

// RobosenseToLidarNode::Callback (ros_wrapping/lidar_conversions/src/RobosenseToLidarNode.cxx:58-125) on the driver's
// organized cloud of pcl::PointXYZI records (height = lasers, width = points per laser, row after row): records with a
// NaN coordinate are skipped (:82-83), a record whose coordinates equal those of the last point KEPT is skipped (dual
// return mode, :87-88), laser_id = i / width through the given mapping, or RS16's when there are 16 lasers (:106-109),
// time from the position inside the ring (:118-119).  layout = {point_step, off_x, off_y, off_z, off_intensity}.
// Returns the number of points kept, -1 for an empty cloud.
int orc_robosense_to_lidar(const void* records, int width, int height, const int32_t layout[5], const uint16_t* mapping, int mapping_len, int device_id, double rpm,
                           lsa_point_t* out)
{
  static const uint16_t LASER_ID_MAPPING_RS16[16] = {0, 1, 2, 3, 4, 5, 6, 7, 15, 14, 13, 12, 11, 10, 9, 8};  // (:32)
  const long long size = (long long)width * height;
  if (size <= 0) return -1;  // "Input RSLidar pointcloud is empty : frame ignored." (:61-65)
  const unsigned char* raw = (const unsigned char*)records;
  const int step = layout[0];
  auto f32 = [&](long long i, int off) { float v; std::memcpy(&v, raw + (size_t)i * step + off, 4); return v; };
  const unsigned int nLasers = (unsigned)height;
  const unsigned int pointsPerRing = (unsigned)(size / nLasers);
  const bool useLaserIdMapping = mapping_len > 0;
  int kept = 0;
  for (unsigned int i = 0; i < (unsigned)size; ++i)
  {
    const float x = f32(i, layout[1]), y = f32(i, layout[2]), z = f32(i, layout[3]);
    if (!(std::isfinite(x) && std::isfinite(y) && std::isfinite(z))) continue;
    if (kept > 0 && x == out[kept - 1].x && y == out[kept - 1].y && z == out[kept - 1].z) continue;  // std::equal on data[0..3)
    lsa_point_t p;
    std::memset(&p, 0, sizeof(p));
    p.x = x; p.y = y; p.z = z; p.w = 1.f;
    p.intensity = f32(i, layout[4]);
    p.device_id = (uint8_t)device_id;
    const uint16_t laser_id = (uint16_t)(i / (unsigned)width);
    p.laser_id = useLaserIdMapping ? mapping[laser_id] : (nLasers == 16) ? LASER_ID_MAPPING_RS16[laser_id] : laser_id;
    const double frameAdvancement = static_cast<double>(i % pointsPerRing) / pointsPerRing;
    p.time = (frameAdvancement - 1) / rpm * 60.;
    out[kept++] = p;
  }
  return kept;
}

// ---- full pipeline ---------------------------------------------------------------------
void* orc_slam_create() { return new SlamHandle; }
void orc_slam_destroy(void* h) { delete (SlamHandle*)h; }
void orc_slam_reset(void* h, int resetLog) { ((SlamHandle*)h)->s.Reset(resetLog != 0); }
int orc_slam_set_param(void* h, const char* name, double v)
{
  Slam& s = ((SlamHandle*)h)->s;
  std::string n(name);
#define P(NAME, EXPR) if (n == NAME) { EXPR; return 0; }
  P("NbThreads", s.NbThreads = (int)v)
  P("UseBlobs", s.UseKeypoints[BLOB] = v != 0)
  P("EgoMotion", s.EgoMotion = (EgoMotionMode)(int)v)
  P("Undistortion", s.Undistortion = (UndistortionMode)(int)v)
  P("TwoDMode", s.TwoDMode = v != 0)
  P("EgoMotionICPMaxIter", s.EgoMotionICPMaxIter = (unsigned)v)
  P("LocalizationICPMaxIter", s.LocalizationICPMaxIter = (unsigned)v)
  P("EgoMotionLMMaxIter", s.EgoMotionLMMaxIter = (unsigned)v)
  P("LocalizationLMMaxIter", s.LocalizationLMMaxIter = (unsigned)v)
  P("EgoMotionMaxNeighborsDistance", s.EgoMotionMaxNeighborsDistance = v)
  P("LocalizationMaxNeighborsDistance", s.LocalizationMaxNeighborsDistance = v)
  P("EgoMotionEdgeNbNeighbors", s.EgoMotionEdgeNbNeighbors = (unsigned)v)
  P("EgoMotionEdgeMinNbNeighbors", s.EgoMotionEdgeMinNbNeighbors = (unsigned)v)
  P("EgoMotionEdgeMaxModelError", s.EgoMotionEdgeMaxModelError = v)
  P("EgoMotionPlaneNbNeighbors", s.EgoMotionPlaneNbNeighbors = (unsigned)v)
  P("EgoMotionPlanarityThreshold", s.EgoMotionPlanarityThreshold = v)
  P("EgoMotionPlaneMaxModelError", s.EgoMotionPlaneMaxModelError = v)
  P("EgoMotionInitSaturationDistance", s.EgoMotionInitSaturationDistance = v)
  P("EgoMotionFinalSaturationDistance", s.EgoMotionFinalSaturationDistance = v)
  P("LocalizationEdgeNbNeighbors", s.LocalizationEdgeNbNeighbors = (unsigned)v)
  P("LocalizationEdgeMinNbNeighbors", s.LocalizationEdgeMinNbNeighbors = (unsigned)v)
  P("LocalizationEdgeMaxModelError", s.LocalizationEdgeMaxModelError = v)
  P("LocalizationPlaneNbNeighbors", s.LocalizationPlaneNbNeighbors = (unsigned)v)
  P("LocalizationPlanarityThreshold", s.LocalizationPlanarityThreshold = v)
  P("LocalizationPlaneMaxModelError", s.LocalizationPlaneMaxModelError = v)
  P("LocalizationBlobNbNeighbors", s.LocalizationBlobNbNeighbors = (unsigned)v)
  P("LocalizationInitSaturationDistance", s.LocalizationInitSaturationDistance = v)
  P("LocalizationFinalSaturationDistance", s.LocalizationFinalSaturationDistance = v)
  P("MaxExtrapolationRatio", s.MaxExtrapolationRatio = v)
  P("MinNbMatchedKeypoints", s.MinNbMatchedKeypoints = (unsigned)v)
  P("KfDistanceThreshold", s.KfDistanceThreshold = v)
  P("KfAngleThreshold", s.KfAngleThreshold = v)
  P("MapUpdate", s.MapUpdate = (MappingMode)(int)v)
  P("OverlapSamplingRatio", s.OverlapSamplingRatio = std::min(std::max((float)v, 0.f), 1.f); if (s.OverlapSamplingRatio == 0.f) s.OverlapEstimation = -1.f)
  P("LoggingTimeout", s.LoggingTimeout = v)
  P("TimeWindowDuration", s.TimeWindowDuration = (float)v)
  P("VelocityLimitLinear", s.VelocityLimits[0] = (float)v)
  P("VelocityLimitAngular", s.VelocityLimits[1] = (float)v)
  P("AccelerationLimitLinear", s.AccelerationLimits[0] = (float)v)
  P("AccelerationLimitAngular", s.AccelerationLimits[1] = (float)v)
  P("Latency", s.Latency = v)
  P("VoxelGridLeafSizeEdges", s.LocalMaps[EDGE]->SetLeafSize(v))
  P("VoxelGridLeafSizePlanes", s.LocalMaps[PLANE]->SetLeafSize(v))
  P("VoxelGridLeafSizeBlobs", s.LocalMaps[BLOB]->SetLeafSize(v))
  P("VoxelGridSize", for (int k = 0; k < 3; ++k) s.LocalMaps[k]->SetGridSize((int)v))
  P("VoxelGridResolution", for (int k = 0; k < 3; ++k) s.LocalMaps[k]->SetVoxelResolution(v))
  P("VoxelGridMinFramesPerVoxel", for (int k = 0; k < 3; ++k) s.LocalMaps[k]->SetMinFramesPerVoxel((unsigned)v))
  P("VoxelGridDecayingThreshold", for (int k = 0; k < 3; ++k) s.LocalMaps[k]->SetDecayingThreshold(v))
  P("VoxelGridSamplingMode", for (int k = 0; k < 3; ++k) s.LocalMaps[k]->SetSampling((SamplingMode)(int)v))
  P("OrderedMaps", for (int k = 0; k < 3; ++k) s.LocalMaps[k]->SetOrdered(v != 0))
  P("NeighborWidth", s.KeyPointsExtractor.P.NeighborWidth = (int)v)
  P("MinDistanceToSensor", s.KeyPointsExtractor.P.MinDistanceToSensor = (float)v)
  P("MinBeamSurfaceAngle", s.KeyPointsExtractor.P.MinBeamSurfaceAngle = (float)v)
  P("PlaneSinAngleThreshold", s.KeyPointsExtractor.P.PlaneSinAngleThreshold = (float)v)
  P("EdgeSinAngleThreshold", s.KeyPointsExtractor.P.EdgeSinAngleThreshold = (float)v)
  P("EdgeDepthGapThreshold", s.KeyPointsExtractor.P.EdgeDepthGapThreshold = (float)v)
  P("EdgeSaliencyThreshold", s.KeyPointsExtractor.P.EdgeSaliencyThreshold = (float)v)
  P("EdgeIntensityGapThreshold", s.KeyPointsExtractor.P.EdgeIntensityGapThreshold = (float)v)
  P("AzimuthalResolution", s.KeyPointsExtractor.AzimuthalResolution = (float)v)
#undef P
  return -3;
}
int orc_slam_add_frame(void* h, const lsa_point_t* pts, int n, uint64_t stampUs, uint32_t seq)
{
  SlamHandle* sh = (SlamHandle*)h;
  std::vector<Point> f((const Point*)pts, (const Point*)pts + n);
  // the frame must outlive the call (GetRegisteredFrame reads it lazily)
  sh->frame.swap(f);
  if (n > 0 && sh->frame[0].device_id != 0 && (!sh->s.OtherExtractors.empty() || !sh->s.OtherBaseToLidarOffsets.empty()))
    sh->s.AddFrames({&sh->frame}, {stampUs});  // a device with an extractor or an offset of its own
  else
    sh->s.AddFrame(sh->frame, stampUs, seq);
  return 0;
}
int orc_slam_get_world_transform(void* h, double T[16], double* time)
{
  Slam& s = ((SlamHandle*)h)->s;
  IsoToRowMajor(s.GetWorldTransform(), T);
  if (time) *time = s.LogTrajectory.empty() ? 0. : s.LogTrajectory.back().time;
  return 0;
}
float orc_slam_get_overlap(void* h) { return ((SlamHandle*)h)->s.OverlapEstimation; }
int orc_slam_get_covariance(void* h, double cov[36])
{
  std::memcpy(cov, ((SlamHandle*)h)->s.GetTransformCovariance(), 36 * sizeof(double));
  return 0;
}
int orc_slam_get_keypoints(void* h, int type, int which, lsa_point_t* out, int capacity)
{
  Slam& s = ((SlamHandle*)h)->s;
  std::vector<Point> world;
  const std::vector<Point>* src;
  if (which == 0) src = &s.CurrentUndistortedKeypoints[type];
  else if (which == 2) src = &s.CurrentRawKeypoints[type];
  else
  {
    world = s.CurrentUndistortedKeypoints[type];
    for (auto& p : world) transform_point(p, s.Tworld);
    src = &world;
  }
  int n = std::min<int>(capacity, src->size());
  std::memcpy(out, src->data(), n * sizeof(Point));
  return n;
}
int orc_slam_get_registered_frame(void* h, lsa_point_t* out, int capacity)
{
  std::vector<Point> f = ((SlamHandle*)h)->s.GetRegisteredFrame();
  int n = std::min<int>(capacity, f.size());
  std::memcpy(out, f.data(), n * sizeof(Point));
  return n;
}
int orc_slam_get_match_status(void* h, int localization, int type, uint8_t* status, double* weights, int capacity)
{
  Slam& s = ((SlamHandle*)h)->s;
  if (!localization && type > 1) return 0;
  const MatchingResults& r = localization ? s.LocalizationMatchingResults[type] : s.EgoMotionMatchingResults[type];
  int n = std::min<int>(capacity, r.Rejections.size());
  for (int i = 0; i < n; ++i) { status[i] = r.Rejections[i]; weights[i] = r.Weights[i]; }
  return n;
}
int orc_slam_get_stats(void* h, double out[16])
{
  Slam& s = ((SlamHandle*)h)->s;
  const StageTimes& t = s.Times;
  double v[16] = {t.total, t.extract, t.ego_icp, t.ego_lm, t.loc_icp, t.loc_lm, t.undistort, t.submap, t.maps,
                  (double)t.ego_iters, (double)t.loc_iters, (double)t.lm_evals, (double)s.TotalMatchedKeypoints, (double)s.KfCounter,
                  0, 0};
  std::memcpy(out, v, sizeof(v));
  return 0;
}
int orc_slam_get_latency_compensated_world_transform(void* h, double T[16])
{
  IsoToRowMajor(((SlamHandle*)h)->s.GetLatencyCompensatedWorldTransform(), T);
  return 0;
}
int orc_slam_set_base_to_lidar_offset(void* h, const double T[16], int device)
{
  Slam& s = ((SlamHandle*)h)->s;
  if (device == 0) s.BaseToLidarOffset = IsoFromRowMajor(T);
  else s.OtherBaseToLidarOffsets[device] = IsoFromRowMajor(T);
  return 0;
}
int orc_slam_set_extractor_param(void* h, int device, const char* name, double v)
{
  Slam& s = ((SlamHandle*)h)->s;
  Extractor& ke = device == 0 ? s.KeyPointsExtractor : s.OtherExtractors[device];
  const std::string n(name);
  if (n == "NeighborWidth") ke.P.NeighborWidth = (int)v;
  else if (n == "MinDistanceToSensor") ke.P.MinDistanceToSensor = (float)v;
  else if (n == "MinBeamSurfaceAngle") ke.P.MinBeamSurfaceAngle = (float)v;
  else if (n == "PlaneSinAngleThreshold") ke.P.PlaneSinAngleThreshold = (float)v;
  else if (n == "EdgeSinAngleThreshold") ke.P.EdgeSinAngleThreshold = (float)v;
  else if (n == "EdgeDepthGapThreshold") ke.P.EdgeDepthGapThreshold = (float)v;
  else if (n == "EdgeSaliencyThreshold") ke.P.EdgeSaliencyThreshold = (float)v;
  else if (n == "EdgeIntensityGapThreshold") ke.P.EdgeIntensityGapThreshold = (float)v;
  else if (n == "AzimuthalResolution") ke.AzimuthalResolution = (float)v;
  else return -3;
  return 0;
}
int orc_slam_add_frames(void* h, const lsa_point_t* const* pts, const int* n, const uint64_t* stampsUs, int nframes)
{
  SlamHandle* sh = (SlamHandle*)h;
  sh->frames.assign(nframes, std::vector<Point>());
  std::vector<const std::vector<Point>*> ptrs;
  std::vector<uint64_t> stamps;
  for (int i = 0; i < nframes; ++i)
  {
    sh->frames[i].resize(n[i]);
    if (n[i] > 0) std::memcpy(sh->frames[i].data(), pts[i], (size_t)n[i] * sizeof(Point));
    ptrs.push_back(&sh->frames[i]);
    stamps.push_back(stampsUs[i]);
  }
  sh->s.AddFrames(ptrs, stamps);
  return 0;
}
int orc_slam_set_world_transform_from_guess(void* h, const double T[16])
{
  ((SlamHandle*)h)->s.SetWorldTransformFromGuess(IsoFromRowMajor(T));
  return 0;
}
int orc_slam_get_trajectory(void* h, double* poses, double* covariances, int capacity)
{
  Slam& s = ((SlamHandle*)h)->s;
  const int n = (int)s.LogTrajectory.size();
  for (int i = 0; i < std::min(n, capacity); ++i)
  {
    if (poses)
    {
      IsoToRowMajor(s.LogTrajectory[i].pose, poses + 17 * i);
      poses[17 * i + 16] = s.LogTrajectory[i].time;
    }
    if (covariances)
    {
      const long j = (long)s.LogCovariances.size() - n + i;
      for (int c = 0; c < 36; ++c) covariances[36 * i + c] = j >= 0 ? s.LogCovariances[j][c] : 0.;
    }
  }
  return n;
}
int orc_slam_get_debug_information(void* h, double out[10])
{
  ((SlamHandle*)h)->s.GetDebugInformation(out);
  return 0;
}
int orc_slam_get_map(void* h, int type, int clean, lsa_point_t* out, int capacity)
{
  const std::vector<Point> m = ((SlamHandle*)h)->s.LocalMaps[type]->Get(clean != 0);
  const int n = (int)m.size();
  if (std::min(n, capacity) > 0) std::memcpy(out, m.data(), (size_t)std::min(n, capacity) * sizeof(Point));
  return n;
}
int orc_slam_get_submap(void* h, int type, lsa_point_t* out, int capacity)
{
  const std::vector<Point>& m = ((SlamHandle*)h)->s.LocalMaps[type]->GetSubMap();
  int n = std::min<int>(capacity, m.size());
  std::memcpy(out, m.data(), n * sizeof(Point));
  return n;
}

// ---- RollingGrid on its own (tests/test_rolling_grid.py) ----
void* orc_grid_create() { return new RollingGrid; }
void orc_grid_destroy(void* g) { delete (RollingGrid*)g; }
int orc_grid_set(void* h, const char* name, double v)
{
  RollingGrid& g = *(RollingGrid*)h;
  const std::string n(name);
  if (n == "GridSize") g.SetGridSize((int)v);
  else if (n == "VoxelResolution") g.SetVoxelResolution(v);
  else if (n == "LeafSize") g.SetLeafSize(v);
  else if (n == "MinFramesPerVoxel") g.SetMinFramesPerVoxel((unsigned)v);
  else if (n == "Sampling") g.SetSampling((SamplingMode)(int)v);
  else if (n == "Ordered") g.SetOrdered(v != 0);
  else if (n == "DecayingThreshold") g.SetDecayingThreshold(v);
  else return -1;
  return 0;
}
void orc_grid_reset(void* g, const float* position) { ((RollingGrid*)g)->Reset(position); }
void orc_grid_clear(void* g) { ((RollingGrid*)g)->Clear(); }
int orc_grid_size(void* g) { return (int)((RollingGrid*)g)->Size(); }
void orc_grid_roll(void* g, const float* mn, const float* mx) { ((RollingGrid*)g)->Roll(mn, mx); }
void orc_grid_add(void* g, const lsa_point_t* pts, int n, int fixed, double time, int roll)
{
  std::vector<Point> pc(n);
  if (n > 0) std::memcpy(pc.data(), pts, (size_t)n * sizeof(Point));
  ((RollingGrid*)g)->Add(pc, fixed != 0, time, roll != 0);
}
void orc_grid_clear_old_points(void* g, double time) { ((RollingGrid*)g)->ClearOldPoints(time); }
int orc_grid_get(void* g, int clean, lsa_point_t* out, int capacity)
{
  const std::vector<Point> pc = ((RollingGrid*)g)->Get(clean != 0);
  const int n = std::min<int>(capacity, pc.size());
  if (n > 0) std::memcpy(out, pc.data(), (size_t)n * sizeof(Point));
  return n;
}
int orc_grid_build_submap(void* h, const float* mn, const float* mx, int minNbPoints)
{
  RollingGrid& g = *(RollingGrid*)h;
  if (mn) g.BuildSubMapKdTree(mn, mx, minNbPoints);
  else g.BuildSubMapKdTree();
  return (int)g.GetSubMap().size();
}
int orc_grid_submap_valid(void* g) { return ((RollingGrid*)g)->IsSubMapKdTreeValid() ? 1 : 0; }
int orc_grid_submap(void* g, lsa_point_t* out, int capacity)
{
  const std::vector<Point>& m = ((RollingGrid*)g)->GetSubMap();
  const int n = std::min<int>(capacity, m.size());
  if (n > 0) std::memcpy(out, m.data(), (size_t)n * sizeof(Point));
  return n;
}

}  // extern "C"
