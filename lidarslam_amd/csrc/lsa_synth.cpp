// Deterministic synthetic spinning-LiDAR sequence generator.
//
// The reference ships no data (ros_wrapping/tests/README.md:4), so tests and
// bench.py feed both the HIP path and the CPU oracle from this generator.
// Input layout mimics what the reference's drivers hand to Slam::AddFrame:
//   * firing order = azimuth-major, rings interleaved within a column
//     (ros_wrapping/lidar_conversions/src/VelodyneToLidarNode.cxx:81-111)
//   * laser_id 0 = lowest ring, increasing upward
//     (slam_lib/include/LidarSlam/SpinningSensorKeypointExtractor.h:84)
//   * per-point `time` = offset <= 0 from the frame stamp, stamp = end of sweep
//     (paraview_wrapping/Plugin/vtkLidarSlam/vtkSlam.cxx:683,697)
// Scene: infinite street canyon along +x -- ground z=-1.8 m, facades y=+-8 m
// with 2 m wide / 1 m deep recesses every 6 m, 0.3 m poles every 10 m at
// y=+-5 m.  Trajectory: 5 m/s along +x with a 3 deg sinusoidal yaw (10 s
// period); every ray is cast from the sensor pose at its own firing time, so
// the sweep carries real motion distortion.  Range noise N(0, 1 cm), continuous
// intensities (no sort ties), rays without a return within 120 m are dropped.
// PRNG: splitmix64 keyed by (seed, frame, point) -- frames are independent.
#include <cmath>
#include <cstdint>
#include <cstring>
#include "../../include/lidarslam_amd.h"

namespace
{
inline uint64_t splitmix64(uint64_t& s)
{
  uint64_t z = (s += 0x9E3779B97F4A7C15ULL);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
  return z ^ (z >> 31);
}
inline double u01(uint64_t& s) { return (double)(splitmix64(s) >> 11) * (1.0 / 9007199254740992.0); }

constexpr double kSpeed = 5.0;                       // m/s
constexpr double kYawAmp = 3.0 * M_PI / 180.0;       // rad
constexpr double kOmega = 2.0 * M_PI / 10.0;         // rad/s
constexpr double kSweep = 0.1;                       // s
constexpr double kMaxRange = 120.0;
constexpr double kGroundZ = -1.8, kWallY = 8.0, kWallTop = 12.0;
constexpr double kRecessPeriod = 6.0, kRecessLo = 2.0, kRecessHi = 4.0, kRecessDepth = 1.0;
constexpr double kPolePeriod = 10.0, kPoleOffset = 5.0, kPoleY = 5.0, kPoleR = 0.3, kPoleTop = 4.0;

struct Pose2 { double x, y, yaw; };
inline Pose2 pose_at(double t)
{
  Pose2 p;
  p.yaw = kYawAmp * std::sin(kOmega * t);
  p.x = kSpeed * t;
  p.y = kSpeed * kYawAmp / kOmega * (1.0 - std::cos(kOmega * t));
  return p;
}

struct Hit { double s; float base_intensity; };

// nearest intersection of the ray o + s d (|d| = 1) with the scene
inline bool cast(const double o[3], const double d[3], Hit& hit)
{
  double best = kMaxRange;
  float mat = 0.f;
  bool found = false;
  // ground
  if (d[2] < -1e-9)
  {
    double s = (kGroundZ - o[2]) / d[2];
    if (s > 0 && s < best) { best = s; mat = 20.f; found = true; }
  }
  // facades
  for (int side = -1; side <= 1; side += 2)
  {
    double dy = d[1] * side;  // motion towards this facade
    if (dy <= 1e-9) continue;
    double oy = o[1] * side;
    double s = (kWallY - oy) / dy;
    if (s <= 0 || s >= best) continue;
    double xh = o[0] + s * d[0];
    double cell = std::floor(xh / kRecessPeriod);
    double xr = xh - cell * kRecessPeriod;
    double sh = s;
    if (xr > kRecessLo && xr < kRecessHi)
    {
      // the ray enters the recess opening: back wall or one of the side walls
      double sb = (kWallY + kRecessDepth - oy) / dy;
      double xb = o[0] + sb * d[0] - cell * kRecessPeriod;
      if (xb >= kRecessLo && xb <= kRecessHi)
        sh = sb;
      else
      {
        double xs = cell * kRecessPeriod + (d[0] > 0 ? kRecessHi : kRecessLo);
        sh = (xs - o[0]) / d[0];
      }
    }
    double zh = o[2] + sh * d[2];
    if (sh > 0 && sh < best && zh <= kWallTop && zh >= kGroundZ) { best = sh; mat = 80.f; found = true; }
  }
  // poles (vertical cylinders); only those the ray can reach near y = +-kPoleY
  double nxy = std::sqrt(d[0] * d[0] + d[1] * d[1]);
  if (nxy > 1e-9)
  {
    double ux = d[0] / nxy, uy = d[1] / nxy;
    for (int side = -1; side <= 1; side += 2)
    {
      double uys = uy * side;
      if (uys <= 1e-6) continue;
      double s5 = (kPoleY - o[1] * side) / uys;  // planar distance to the pole line
      if (s5 <= 0 || s5 / 1.0 > kMaxRange) continue;
      double x5 = o[0] + s5 * ux;
      double half = kPoleR / uys + 1e-6;
      int k0 = (int)std::floor((x5 - half - kPoleOffset) / kPolePeriod);
      int k1 = (int)std::ceil((x5 + half - kPoleOffset) / kPolePeriod);
      if (k1 - k0 > 8) k1 = k0 + 8;
      for (int k = k0; k <= k1; ++k)
      {
        double cx = k * kPolePeriod + kPoleOffset, cy = kPoleY * side;
        double rx = o[0] - cx, ry = o[1] - cy;
        double b = rx * ux + ry * uy;
        double c = rx * rx + ry * ry - kPoleR * kPoleR;
        double disc = b * b - c;
        if (disc < 0) continue;
        double sp = -b - std::sqrt(disc);  // planar
        if (sp <= 0) continue;
        double s = sp / nxy;
        double zh = o[2] + s * d[2];
        if (s < best && zh <= kPoleTop && zh >= kGroundZ) { best = s; mat = 200.f; found = true; }
      }
    }
  }
  hit.s = best;
  hit.base_intensity = mat;
  return found;
}
}  // namespace

extern "C" int lsa_synth_sensor(int model, int* nrings, int* ncols, double* el_min_deg, double* el_max_deg)
{
  switch (model)
  {
    case 8: *nrings = 8; *ncols = 400; *el_min_deg = -16.0; *el_max_deg = 12.0; return 0;  // miniature sensor for fixtures
    case 16: *nrings = 16; *ncols = 1800; *el_min_deg = -15.0; *el_max_deg = 15.0; return 0;
    case 64: *nrings = 64; *ncols = 2048; *el_min_deg = -24.8; *el_max_deg = 2.0; return 0;
    case 128: *nrings = 128; *ncols = 2048; *el_min_deg = -25.0; *el_max_deg = 15.0; return 0;
    default: return -1;
  }
}

extern "C" void lsa_synth_pose(int frame, double T[16])
{
  // pose of BASE at the stamp of `frame`, relative to the pose at the stamp of frame 0
  Pose2 p0 = pose_at(kSweep * 1), p = pose_at(kSweep * (frame + 1));
  double c0 = std::cos(p0.yaw), s0 = std::sin(p0.yaw);
  double dx = p.x - p0.x, dy = p.y - p0.y;
  double yaw = p.yaw - p0.yaw;
  double c = std::cos(yaw), s = std::sin(yaw);
  double M[16] = {c, -s, 0, c0 * dx + s0 * dy, s, c, 0, -s0 * dx + c0 * dy, 0, 0, 1, 0, 0, 0, 0, 1};
  std::memcpy(T, M, sizeof(M));
}

extern "C" int lsa_synth_frame(int model, uint64_t seed, int frame, lsa_point_t* out, int capacity, uint64_t* stamp_us)
{
  int nrings, ncols;
  double elmin, elmax;
  if (lsa_synth_sensor(model, &nrings, &ncols, &elmin, &elmax) != 0) return -1;
  const double t_end = kSweep * (frame + 1);
  if (stamp_us) *stamp_us = (uint64_t)std::llround(t_end * 1e6);
  int n = 0;
  for (int c = 0; c < ncols; ++c)
  {
    const double toff = ((double)c / ncols - 1.0) * kSweep;
    const Pose2 p = pose_at(t_end + toff);
    const double cy = std::cos(p.yaw), sy = std::sin(p.yaw);
    const double az = 2.0 * M_PI * c / ncols;
    const double ca = std::cos(az), sa = std::sin(az);
    for (int r = 0; r < nrings; ++r)
    {
      uint64_t s = seed * 0x9E3779B97F4A7C15ULL + ((uint64_t)frame << 32) + (uint64_t)(c * nrings + r);
      splitmix64(s);
      const double el = (elmin + (elmax - elmin) * r / (nrings - 1)) * M_PI / 180.0;
      const double ce = std::cos(el), se = std::sin(el);
      const double ds[3] = {ce * ca, -ce * sa, se};  // sensor frame, clockwise spin
      const double d[3] = {cy * ds[0] - sy * ds[1], sy * ds[0] + cy * ds[1], ds[2]};
      const double o[3] = {p.x, p.y, 0.0};
      Hit h;
      // draw the random numbers unconditionally so that they do not depend on the geometry
      double u1 = u01(s), u2 = u01(s), u3 = u01(s);
      if (!cast(o, d, h)) continue;
      double noise = 0.01 * std::sqrt(-2.0 * std::log(u1 + 1e-300)) * std::cos(2.0 * M_PI * u2);
      double range = h.s + noise;
      if (range >= kMaxRange || range <= 0.05) continue;
      if (n >= capacity) return -2;
      lsa_point_t q;
      std::memset(&q, 0, sizeof(q));
      q.x = (float)(range * ds[0]);
      q.y = (float)(range * ds[1]);
      q.z = (float)(range * ds[2]);
      q.w = 1.0f;
      q.time = toff;
      q.intensity = (float)(h.base_intensity + 10.0 * (2.0 * u3 - 1.0));
      q.laser_id = (uint16_t)r;
      q.device_id = 0;
      q.label = 0;
      out[n++] = q;
    }
  }
  return n;
}
