// lsa_hostmath.h -- host-side pose algebra of the product (6-dof control flow stays on the
// host: it is a handful of 4x4 products per ICP iteration).  Restates the Eigen operations the
// reference's Slam.cxx / Utilities.cxx / MotionModel.cxx perform on Eigen::Isometry3d:
//   Utils::XYZRPYtoIsometry / IsometryToXYZRPY   slam_lib/src/Utilities.cxx:33-77
//   LinearInterpolation                           slam_lib/src/MotionModel.cxx:26-34
//   LinearTransformInterpolator<double>           slam_lib/include/LidarSlam/MotionModel.h:36-136
#pragma once
#include <algorithm>
#include <cmath>
#include <cstring>
#include <limits>
#include "../../../include/lsa_pmath.h"

namespace lsa
{
namespace host
{

// rigid transform stored as a row-major 4x4 (the layout of the C ABI)
struct Pose
{
  double m[16];
  static Pose Identity()
  {
    Pose p;
    for (int i = 0; i < 16; ++i) p.m[i] = (i % 5 == 0) ? 1. : 0.;
    return p;
  }
  double& operator()(int r, int c) { return m[r * 4 + c]; }
  double operator()(int r, int c) const { return m[r * 4 + c]; }
};

inline Pose operator*(const Pose& a, const Pose& b)
{
  Pose r = Pose::Identity();
  for (int i = 0; i < 3; ++i)
  {
    for (int j = 0; j < 3; ++j) r(i, j) = (a(i, 0) * b(0, j) + a(i, 1) * b(1, j)) + a(i, 2) * b(2, j);
    r(i, 3) = ((a(i, 0) * b(0, 3) + a(i, 1) * b(1, 3)) + a(i, 2) * b(2, 3)) + a(i, 3);
  }
  return r;
}
inline Pose Inverse(const Pose& a)
{
  Pose r = Pose::Identity();
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) r(i, j) = a(j, i);
  for (int i = 0; i < 3; ++i) r(i, 3) = -((r(i, 0) * a(0, 3) + r(i, 1) * a(1, 3)) + r(i, 2) * a(2, 3));
  return r;
}
// Eigen isApprox on the 4x4, precision 1e-12
inline bool IsApprox(const Pose& a, const Pose& b)
{
  double d = 0, na = 0, nb = 0;
  for (int i = 0; i < 16; ++i)
  {
    const double e = a.m[i] - b.m[i];
    d += e * e; na += a.m[i] * a.m[i]; nb += b.m[i] * b.m[i];
  }
  return d <= 1e-24 * std::min(na, nb);
}

struct Quaternion
{
  double w, x, y, z;
};
inline Quaternion ToQuaternion(const Pose& p)
{
  Quaternion q;
  double t = p(0, 0) + p(1, 1) + p(2, 2);
  if (t > 0.0)
  {
    t = std::sqrt(t + 1.0);
    q.w = 0.5 * t;
    t = 0.5 / t;
    q.x = (p(2, 1) - p(1, 2)) * t;
    q.y = (p(0, 2) - p(2, 0)) * t;
    q.z = (p(1, 0) - p(0, 1)) * t;
  }
  else
  {
    int i = 0;
    if (p(1, 1) > p(0, 0)) i = 1;
    if (p(2, 2) > p(i, i)) i = 2;
    const int j = (i + 1) % 3, k = (j + 1) % 3;
    t = std::sqrt(p(i, i) - p(j, j) - p(k, k) + 1.0);
    double v[3];
    v[i] = 0.5 * t;
    t = 0.5 / t;
    q.w = (p(k, j) - p(j, k)) * t;
    v[j] = (p(j, i) + p(i, j)) * t;
    v[k] = (p(k, i) + p(i, k)) * t;
    q.x = v[0]; q.y = v[1]; q.z = v[2];
  }
  return q;
}
inline void SetRotation(Pose& p, const Quaternion& q)
{
  const double tx = 2.0 * q.x, ty = 2.0 * q.y, tz = 2.0 * q.z;
  const double twx = tx * q.w, twy = ty * q.w, twz = tz * q.w;
  const double txx = tx * q.x, txy = ty * q.x, txz = tz * q.x;
  const double tyy = ty * q.y, tyz = tz * q.y, tzz = tz * q.z;
  p(0, 0) = 1.0 - (tyy + tzz); p(0, 1) = txy - twz;         p(0, 2) = txz + twy;
  p(1, 0) = txy + twz;         p(1, 1) = 1.0 - (txx + tzz); p(1, 2) = tyz - twx;
  p(2, 0) = txz - twy;         p(2, 1) = tyz + twx;         p(2, 2) = 1.0 - (txx + tyy);
}
inline Quaternion Mul(const Quaternion& a, const Quaternion& b)
{
  return {a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z, a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y,
          a.w * b.y + a.y * b.w + a.z * b.x - a.x * b.z, a.w * b.z + a.z * b.w + a.x * b.y - a.y * b.x};
}
// Eigen::Quaterniond::slerp; sines through lsa_pmath (the device evaluates the same expression
// per point, so the host uses the same routine)
inline Quaternion Slerp(const Quaternion& a, const Quaternion& b, double t)
{
  const double d = (a.x * b.x + a.z * b.z) + (a.y * b.y + a.w * b.w);
  const double absD = std::abs(d);
  double s0, s1;
  if (absD >= 1.0 - std::numeric_limits<double>::epsilon()) { s0 = 1.0 - t; s1 = t; }
  else
  {
    const double theta = std::acos(absD);
    const double sinTheta = lsa_sin(theta);
    s0 = lsa_sin((1.0 - t) * theta) / sinTheta;
    s1 = lsa_sin(t * theta) / sinTheta;
  }
  if (d < 0.0) s1 = -s1;
  return {s0 * a.w + s1 * b.w, s0 * a.x + s1 * b.x, s0 * a.y + s1 * b.y, s0 * a.z + s1 * b.z};
}

// Utils::RPYtoRotationMatrix via AngleAxis products = quaternion products (Utilities.cxx:33-38)
inline Pose FromXYZRPY(const double w[6])
{
  const Quaternion qz = {std::cos(w[5] * 0.5), 0, 0, std::sin(w[5] * 0.5)};
  const Quaternion qy = {std::cos(w[4] * 0.5), 0, std::sin(w[4] * 0.5), 0};
  const Quaternion qx = {std::cos(w[3] * 0.5), std::sin(w[3] * 0.5), 0, 0};
  Pose p = Pose::Identity();
  SetRotation(p, Mul(Mul(qz, qy), qx));
  p(0, 3) = w[0]; p(1, 3) = w[1]; p(2, 3) = w[2];
  return p;
}
// Utils::IsometryToXYZRPY (Utilities.cxx:41-77)
inline void ToXYZRPY(const Pose& p, double w[6])
{
  w[0] = p(0, 3); w[1] = p(1, 3); w[2] = p(2, 3);
  w[3] = std::atan2(p(2, 1), p(2, 2));
  w[4] = -std::asin(p(2, 0));
  w[5] = std::atan2(p(1, 0), p(0, 0));
}

inline Pose LinearInterpolation(const Pose& H0, const Pose& H1, double t, double t0, double t1)
{
  if (t0 == t1 || IsApprox(H0, H1)) return H1;
  const double time = (t - t0) / (t1 - t0);
  Pose r = Pose::Identity();
  SetRotation(r, Slerp(ToQuaternion(H0), ToQuaternion(H1), time));
  for (int i = 0; i < 3; ++i) r(i, 3) = H0(i, 3) + time * (H1(i, 3) - H0(i, 3));
  return r;
}

// rotation angle of Eigen::AngleAxisd(R)
inline double RotationAngle(const Pose& p)
{
  const Quaternion q = ToQuaternion(p);
  const double n = std::sqrt((q.x * q.x + q.y * q.y) + q.z * q.z);
  return (n != 0.) ? 2. * std::atan2(n, std::abs(q.w)) : 0.;
}

// State of LinearTransformInterpolator<double>: rotations kept as quaternions
struct WithinFrameMotion
{
  double Time0 = 0., Time1 = 1.;
  Quaternion Rot0{1, 0, 0, 0}, Rot1{1, 0, 0, 0};
  double Trans0[3] = {0, 0, 0}, Trans1[3] = {0, 0, 0};
  Pose GetH0() const
  {
    Pose p = Pose::Identity();
    SetRotation(p, Rot0);
    for (int i = 0; i < 3; ++i) p(i, 3) = Trans0[i];
    return p;
  }
  Pose GetH1() const
  {
    Pose p = Pose::Identity();
    SetRotation(p, Rot1);
    for (int i = 0; i < 3; ++i) p(i, 3) = Trans1[i];
    return p;
  }
  void SetTransforms(const Pose& H0, const Pose& H1)
  {
    Rot0 = ToQuaternion(H0);
    Rot1 = ToQuaternion(H1);
    for (int i = 0; i < 3; ++i) { Trans0[i] = H0(i, 3); Trans1[i] = H1(i, 3); }
  }
  void SetTimes(double t0, double t1) { Time0 = t0; Time1 = t1; }
  double GetTimeRange() const { return Time1 - Time0; }
};

}  // namespace host
}  // namespace lsa
