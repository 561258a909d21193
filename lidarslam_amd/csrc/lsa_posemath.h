// lsa_posemath.h -- the 6-dof pose algebra between two ICP iterations, ONE source for the host (the loops of
// host/lsa_slam_core.cpp) and the device (the link a solve leaves for the iteration enqueued behind it, lsa_lm.hip); the
// elementary functions are include/lsa_pmath.h's, which the tests' CPU restatement evaluates as well.
// Restates the Eigen operations the reference performs on Eigen::Isometry3d:
//   Utils::XYZRPYtoIsometry / IsometryToXYZRPY   slam_lib/src/Utilities.cxx:33-77
//   LinearInterpolation                           slam_lib/src/MotionModel.cxx:26-34
//   LinearTransformInterpolator<double>           slam_lib/include/LidarSlam/MotionModel.h:36-136
//   Slam::InterpolateScanPose, RefineUndistortion slam_lib/src/Slam.cxx:1271-1285, 1322-1352
// Plain C++ (no HIP header, no libm): sines, cosines and their inverses come from lsa_pmath.h, sqrt is the IEEE one
// on both sides, nothing may be contracted or re-associated (-ffp-contract=off) -- the same inputs give the same bits
// wherever this is compiled (tests/test_gpu_match.py::test_links_left_by_the_device_equal_the_host_algebra).
#pragma once
#include "../../include/lsa_pmath.h"

#if defined(__HIPCC__)
#define LSA_HDM __host__ __device__
#else
#define LSA_HDM
#endif

namespace lsa
{

// double-precision rigid transform: row-major R + t, applied as ((R0 x + R1 y) + R2 z) + t  (Eigen 4x4 * 4x1)
struct Rigid
{
  double R[9];
  double t[3];
};

// Everything of LinearTransformInterpolator::operator() that does not depend on the point's time
struct InterpConst
{
  double qa[4], qb[4];   // w x y z
  double d, theta, sin_theta;
  double trans0[3], trans1[3];
  double time0, time1;
  Rigid h0;              // applied to every point when invalid
  int linear;            // |d| >= 1 - eps
  int invalid;           // Time0 == Time1 or H0 ~ H1
};

// What an ICP iteration that was enqueued ahead of its inputs reads on the device once they are there -- handed over by
// the host through a gate (lsa_icp_gate) or left by the solve in front of it (a link, lsa_icp_link): the pose the keypoints
// are searched under, the optimiser's start point, the undistortion of the iteration before.  `go` comes first: 1 = run,
// anything else = the launch does nothing (0 called off, 2 the gate gave up).
struct IcpInputs
{
  Rigid pose;
  double x0[6];
  InterpConst ic;
};
struct IcpGate
{
  unsigned long long go;
  IcpInputs in;
};
static_assert(sizeof(IcpGate) % 8 == 0 && sizeof(IcpGate) <= 64 * 8, "a gate block is at most 64 words");

namespace posemath
{

// rigid transform stored as a row-major 4x4 (the layout of the C ABI)
struct Pose
{
  double m[16];
  LSA_HDM static Pose Identity()
  {
    Pose p;
    for (int i = 0; i < 16; ++i) p.m[i] = (i % 5 == 0) ? 1. : 0.;
    return p;
  }
  LSA_HDM double& operator()(int r, int c) { return m[r * 4 + c]; }
  LSA_HDM double operator()(int r, int c) const { return m[r * 4 + c]; }
};

LSA_HD Pose operator*(const Pose& a, const Pose& b)
{
  Pose r = Pose::Identity();
  for (int i = 0; i < 3; ++i)
  {
    for (int j = 0; j < 3; ++j) r(i, j) = (a(i, 0) * b(0, j) + a(i, 1) * b(1, j)) + a(i, 2) * b(2, j);
    r(i, 3) = ((a(i, 0) * b(0, 3) + a(i, 1) * b(1, 3)) + a(i, 2) * b(2, 3)) + a(i, 3);
  }
  return r;
}
LSA_HD Pose Inverse(const Pose& a)
{
  Pose r = Pose::Identity();
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) r(i, j) = a(j, i);
  for (int i = 0; i < 3; ++i) r(i, 3) = -((r(i, 0) * a(0, 3) + r(i, 1) * a(1, 3)) + r(i, 2) * a(2, 3));
  return r;
}
// Eigen isApprox on the 4x4, precision 1e-12
LSA_HD bool IsApprox(const Pose& a, const Pose& b)
{
  double d = 0, na = 0, nb = 0;
  for (int i = 0; i < 16; ++i)
  {
    const double e = a.m[i] - b.m[i];
    d += e * e; na += a.m[i] * a.m[i]; nb += b.m[i] * b.m[i];
  }
  return d <= 1e-24 * (nb < na ? nb : na);  // std::min(na, nb)
}
LSA_HD Rigid ToRigid(const Pose& p)
{
  Rigid r;
  for (int i = 0; i < 3; ++i)
  {
    for (int j = 0; j < 3; ++j) r.R[i * 3 + j] = p(i, j);
    r.t[i] = p(i, 3);
  }
  return r;
}
LSA_HD Pose FromRigid(const Rigid& r)
{
  Pose p = Pose::Identity();
  for (int i = 0; i < 3; ++i)
  {
    for (int j = 0; j < 3; ++j) p(i, j) = r.R[i * 3 + j];
    p(i, 3) = r.t[i];
  }
  return p;
}

struct Quaternion
{
  double w, x, y, z;
};
// Eigen::Quaternion(Matrix3d), the branch for a trace <= 0 with I the largest diagonal element (constant indices only:
// nothing of a pose is addressed by a run-time index, which on the device would put it in scratch memory)
template <int I> LSA_HD Quaternion ToQuaternionLargest(const Pose& p)
{
  constexpr int J = (I + 1) % 3, K = (J + 1) % 3;
  Quaternion q;
  double t = __builtin_sqrt(p(I, I) - p(J, J) - p(K, K) + 1.0);
  const double vi = 0.5 * t;
  t = 0.5 / t;
  q.w = (p(K, J) - p(J, K)) * t;
  const double vj = (p(J, I) + p(I, J)) * t;
  const double vk = (p(K, I) + p(I, K)) * t;
  q.x = I == 0 ? vi : (J == 0 ? vj : vk);
  q.y = I == 1 ? vi : (J == 1 ? vj : vk);
  q.z = I == 2 ? vi : (J == 2 ? vj : vk);
  return q;
}
LSA_HD Quaternion ToQuaternion(const Pose& p)
{
  Quaternion q;
  double t = p(0, 0) + p(1, 1) + p(2, 2);
  if (t > 0.0)
  {
    t = __builtin_sqrt(t + 1.0);
    q.w = 0.5 * t;
    t = 0.5 / t;
    q.x = (p(2, 1) - p(1, 2)) * t;
    q.y = (p(0, 2) - p(2, 0)) * t;
    q.z = (p(1, 0) - p(0, 1)) * t;
    return q;
  }
  int i = 0;
  if (p(1, 1) > p(0, 0)) i = 1;
  if (p(2, 2) > (i == 0 ? p(0, 0) : p(1, 1))) i = 2;
  return i == 0 ? ToQuaternionLargest<0>(p) : (i == 1 ? ToQuaternionLargest<1>(p) : ToQuaternionLargest<2>(p));
}
LSA_HD void SetRotation(Pose& p, const Quaternion& q)
{
  const double tx = 2.0 * q.x, ty = 2.0 * q.y, tz = 2.0 * q.z;
  const double twx = tx * q.w, twy = ty * q.w, twz = tz * q.w;
  const double txx = tx * q.x, txy = ty * q.x, txz = tz * q.x;
  const double tyy = ty * q.y, tyz = tz * q.y, tzz = tz * q.z;
  p(0, 0) = 1.0 - (tyy + tzz); p(0, 1) = txy - twz;         p(0, 2) = txz + twy;
  p(1, 0) = txy + twz;         p(1, 1) = 1.0 - (txx + tzz); p(1, 2) = tyz - twx;
  p(2, 0) = txz - twy;         p(2, 1) = tyz + twx;         p(2, 2) = 1.0 - (txx + tyy);
}
LSA_HD Quaternion Mul(const Quaternion& a, const Quaternion& b)
{
  return {a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z, a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y,
          a.w * b.y + a.y * b.w + a.z * b.x - a.x * b.z, a.w * b.z + a.z * b.w + a.x * b.y - a.y * b.x};
}
constexpr double kOneMinusEps = 1.0 - 2.220446049250313080847e-16;  // 1 - std::numeric_limits<double>::epsilon()
// Eigen::Quaterniond::slerp
LSA_HD Quaternion Slerp(const Quaternion& a, const Quaternion& b, double t)
{
  const double d = (a.x * b.x + a.z * b.z) + (a.y * b.y + a.w * b.w);
  const double absD = __builtin_fabs(d);
  double s0, s1;
  if (absD >= kOneMinusEps) { s0 = 1.0 - t; s1 = t; }
  else
  {
    const double theta = lsa_acos(absD);
    const double sinTheta = lsa_sin(theta);
    s0 = lsa_sin((1.0 - t) * theta) / sinTheta;
    s1 = lsa_sin(t * theta) / sinTheta;
  }
  if (d < 0.0) s1 = -s1;
  return {s0 * a.w + s1 * b.w, s0 * a.x + s1 * b.x, s0 * a.y + s1 * b.y, s0 * a.z + s1 * b.z};
}

// Utils::RPYtoRotationMatrix via AngleAxis products = quaternion products (Utilities.cxx:33-38); the cosines and sines
// of the three half angles may come from elsewhere (six lanes of a wavefront work them out side by side)
LSA_HD Pose FromXYZRPYTrig(const double w[6], double cx, double sx, double cy, double sy, double cz, double sz)
{
  const Quaternion qz = {cz, 0, 0, sz};
  const Quaternion qy = {cy, 0, sy, 0};
  const Quaternion qx = {cx, sx, 0, 0};
  Pose p = Pose::Identity();
  SetRotation(p, Mul(Mul(qz, qy), qx));
  p(0, 3) = w[0]; p(1, 3) = w[1]; p(2, 3) = w[2];
  return p;
}
LSA_HD Pose FromXYZRPY(const double w[6])
{
  return FromXYZRPYTrig(w, lsa_cos(w[3] * 0.5), lsa_sin(w[3] * 0.5), lsa_cos(w[4] * 0.5), lsa_sin(w[4] * 0.5), lsa_cos(w[5] * 0.5), lsa_sin(w[5] * 0.5));
}
// Utils::IsometryToXYZRPY (Utilities.cxx:41-77)
LSA_HD void ToXYZRPY(const Pose& p, double w[6])
{
  w[0] = p(0, 3); w[1] = p(1, 3); w[2] = p(2, 3);
  w[3] = lsa_atan2(p(2, 1), p(2, 2));
  w[4] = -lsa_asin(p(2, 0));
  w[5] = lsa_atan2(p(1, 0), p(0, 0));
}

LSA_HD Pose LinearInterpolation(const Pose& H0, const Pose& H1, double t, double t0, double t1)
{
  if (t0 == t1 || IsApprox(H0, H1)) return H1;
  const double time = (t - t0) / (t1 - t0);
  Pose r = Pose::Identity();
  SetRotation(r, Slerp(ToQuaternion(H0), ToQuaternion(H1), time));
  for (int i = 0; i < 3; ++i) r(i, 3) = H0(i, 3) + time * (H1(i, 3) - H0(i, 3));
  return r;
}

// rotation angle of Eigen::AngleAxisd(R)
LSA_HD double RotationAngle(const Pose& p)
{
  const Quaternion q = ToQuaternion(p);
  const double n = __builtin_sqrt((q.x * q.x + q.y * q.y) + q.z * q.z);
  return (n != 0.) ? 2. * lsa_atan2(n, __builtin_fabs(q.w)) : 0.;
}

// State of LinearTransformInterpolator<double>: rotations kept as quaternions
struct WithinFrameMotion
{
  double Time0 = 0., Time1 = 1.;
  Quaternion Rot0{1, 0, 0, 0}, Rot1{1, 0, 0, 0};
  double Trans0[3] = {0, 0, 0}, Trans1[3] = {0, 0, 0};
  LSA_HDM Pose GetH0() const
  {
    Pose p = Pose::Identity();
    SetRotation(p, Rot0);
    for (int i = 0; i < 3; ++i) p(i, 3) = Trans0[i];
    return p;
  }
  LSA_HDM Pose GetH1() const
  {
    Pose p = Pose::Identity();
    SetRotation(p, Rot1);
    for (int i = 0; i < 3; ++i) p(i, 3) = Trans1[i];
    return p;
  }
  LSA_HDM void SetTransforms(const Pose& H0, const Pose& H1)
  {
    Rot0 = ToQuaternion(H0);
    Rot1 = ToQuaternion(H1);
    for (int i = 0; i < 3; ++i) { Trans0[i] = H0(i, 3); Trans1[i] = H1(i, 3); }
  }
  LSA_HDM void SetTimes(double t0, double t1) { Time0 = t0; Time1 = t1; }
  LSA_HDM double GetTimeRange() const { return Time1 - Time0; }
};

// Everything of LinearTransformInterpolator that is independent of the point: SetTransforms (quaternion round trip),
// IsInterpolatorValid (isApprox, precision 1e-12), the slerp's constants (MotionModel.h:64-129)
LSA_HD InterpConst MakeInterpConst(const Pose& H0, const Pose& H1, double t0, double t1)
{
  InterpConst c;
  for (int i = 0; i < 3; ++i) { c.trans0[i] = H0(i, 3); c.trans1[i] = H1(i, 3); }
  const Quaternion qa = ToQuaternion(H0), qb = ToQuaternion(H1);
  c.qa[0] = qa.w; c.qa[1] = qa.x; c.qa[2] = qa.y; c.qa[3] = qa.z;
  c.qb[0] = qb.w; c.qb[1] = qb.x; c.qb[2] = qb.y; c.qb[3] = qb.z;
  c.time0 = t0; c.time1 = t1;
  // GetH0 / GetH1 go through the quaternions
  Pose G0 = Pose::Identity(), G1 = Pose::Identity();
  SetRotation(G0, qa);
  SetRotation(G1, qb);
  for (int i = 0; i < 3; ++i)
  {
    for (int j = 0; j < 3; ++j) c.h0.R[i * 3 + j] = G0(i, j);
    c.h0.t[i] = c.trans0[i];
  }
  double d = 0, na = 1, nb = 1;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) { const double e = G0(i, j) - G1(i, j); d += e * e; na += G0(i, j) * G0(i, j); nb += G1(i, j) * G1(i, j); }
  for (int i = 0; i < 3; ++i) { const double e = c.trans0[i] - c.trans1[i]; d += e * e; na += c.trans0[i] * c.trans0[i]; nb += c.trans1[i] * c.trans1[i]; }
  const bool approx = d <= 1e-12 * 1e-12 * (nb < na ? nb : na);
  c.invalid = (t0 == t1 || approx) ? 1 : 0;
  c.d = (c.qa[1] * c.qb[1] + c.qa[3] * c.qb[3]) + (c.qa[2] * c.qb[2] + c.qa[0] * c.qb[0]);
  const double absD = __builtin_fabs(c.d);
  c.linear = absD >= kOneMinusEps ? 1 : 0;
  c.theta = c.linear ? 0.0 : lsa_acos(absD);
  c.sin_theta = c.linear ? 1.0 : lsa_sin(c.theta);
  return c;
}

// What Slam::InterpolateScanPose needs beside the two poses (Slam.cxx:1271-1285)
struct ScanPoseClock
{
  int have_log;       // LogTrajectory not empty
  double prev_time;   // LogTrajectory.back().time
  double cur_time;    // the current frame's stamp [s]
  double max_ratio;   // MaxExtrapolationRatio
};
LSA_HD Pose InterpolateScanPose(const ScanPoseClock& k, const Pose& previousTworld, const Pose& tworld, double time)
{
  if (!k.have_log) return tworld;
  if (__builtin_fabs(time / (k.cur_time - k.prev_time)) > k.max_ratio) return tworld;
  return LinearInterpolation(previousTworld, tworld, k.cur_time + time, k.prev_time, k.cur_time);
}
// Slam::RefineUndistortion (Slam.cxx:1322-1352): the motion within the frame under the new pose; d0 / d1 move the
// keypoints undistorted under the old motion to where the new one puts them
// (the scan's poses at its begin and its end handed in: the device interpolates the two on two lanes side by side)
LSA_HD void RefineUndistortionFrom(WithinFrameMotion& motion, const Pose& worldToBaseBegin, const Pose& worldToBaseEnd, const Pose& tworld, Pose& d0, Pose& d1)
{
  const Pose previousBaseBegin = motion.GetH0();
  const Pose previousBaseEnd = motion.GetH1();
  const Pose baseToWorld = Inverse(tworld);
  const Pose newBaseBegin = baseToWorld * worldToBaseBegin;
  const Pose newBaseEnd = baseToWorld * worldToBaseEnd;
  motion.SetTransforms(newBaseBegin, newBaseEnd);
  d0 = newBaseBegin * Inverse(previousBaseBegin);
  d1 = newBaseEnd * Inverse(previousBaseEnd);
}
LSA_HD void RefineUndistortion(WithinFrameMotion& motion, const ScanPoseClock& k, const Pose& previousTworld, const Pose& tworld, Pose& d0, Pose& d1)
{
  const Pose worldToBaseBegin = InterpolateScanPose(k, previousTworld, tworld, motion.Time0);
  const Pose worldToBaseEnd = InterpolateScanPose(k, previousTworld, tworld, motion.Time1);
  RefineUndistortionFrom(motion, worldToBaseBegin, worldToBaseEnd, tworld, d0, d1);
}

}  // namespace posemath
}  // namespace lsa
