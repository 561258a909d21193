// =============================================================================
// ORACLE -- TEST INFRASTRUCTURE ONLY.  Not part of the shipped product.
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
// build, link or call anything under oracle/.
//
// PARITY UNPINNED: the reference (Perception4D/LidarSlam v1.5) holds no golden
// vectors for this path and cannot be built here (Eigen/PCL/Ceres/nanoflann
// absent).  This is a CPU restatement written from the reference sources and
// from the published algorithms of its third-party dependencies.
// =============================================================================
//
// orc_math.hpp -- small fixed-size linear algebra that restates the Eigen 3.3 /
// PCL 1.10 arithmetic the reference reaches on its hot path:
//   * Eigen unrolled reductions (squaredNorm/dot of 3-vectors):
//       float : x + (y + z)   (redux_novec_unroller halving, Size 3 < Packet4f)
//       double: (x + y) + z   (Packet2d + scalar tail)
//   * Eigen::MatrixBase::normalized()  (guards z > 0)
//   * pcl::computeMeanAndCovarianceMatrix (PCL 1.10: products in float, then
//     accumulated in Scalar; single pass, 1/N normalisation)
//     -- slam_lib/include/LidarSlam/Utilities.h:257
//   * pcl::eigen33 / computeRoots / computeRoots2 (analytic symmetric 3x3)
//     -- slam_lib/include/LidarSlam/Utilities.h:261
//   * Eigen quaternion <-> matrix, slerp, AngleAxis products, Isometry algebra
//     -- slam_lib/src/Utilities.cxx:33-77, slam_lib/src/MotionModel.cxx:26-34,
//        slam_lib/include/LidarSlam/MotionModel.h:36-136
// Transcendentals that the GPU path also evaluates (atan2/cos/sin inside
// eigen33, sin inside slerp) come from include/lsa_pmath.h so that CPU and GPU
// agree bit for bit -- since round 3 also the pose algebra between two ICP iterations (RPY conversion, acos in slerp),
// which the device evaluates behind a solve when the next iteration is enqueued ahead (lsa_posemath.h).
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <algorithm>
#include "../include/lsa_pmath.h"

namespace orc
{
// The trigonometry inside the analytic eigen-solver, the slerp and the RPY conversions: lsa_pmath.h (what the device evaluates, bit for bit)
// by default; libm (what the reference's PCL / Eigen call: std::atan2 / cos / sin of glibc) when asked for.  The
// switch exists for ONE purpose: tests/test_oracle_properties.py counts how many decisions (keypoint labels, match
// status) and how many low bits change between the two, i.e. what the shared header could hide.
inline int& libm_trig() { static int on = 0; return on; }
inline double t_sin(double x) { return libm_trig() ? std::sin(x) : lsa_sin(x); }
inline double t_cos(double x) { return libm_trig() ? std::cos(x) : lsa_cos(x); }
inline double t_atan2(double y, double x) { return libm_trig() ? std::atan2(y, x) : lsa_atan2(y, x); }
inline double t_asin(double x) { return libm_trig() ? std::asin(x) : lsa_asin(x); }
inline double t_acos(double x) { return libm_trig() ? std::acos(x) : lsa_acos(x); }


// ---------------------------------------------------------------------------
// 32-byte point, same layout as LidarSlam::LidarPoint
// (slam_lib/include/LidarSlam/LidarPoint.h:31-64)
struct Point
{
  float x, y, z, w;
  double time;
  float intensity;
  uint16_t laser_id;
  uint8_t device_id;
  uint8_t label;
};
static_assert(sizeof(Point) == 32, "LidarPoint must be 32 bytes");

// ---------------------------------------------------------------------------
template <typename T> struct V3 { T x, y, z; };
using V3f = V3<float>;
using V3d = V3<double>;

template <typename T> inline T sum3(T a, T b, T c);
template <> inline float sum3<float>(float a, float b, float c) { return a + (b + c); }
template <> inline double sum3<double>(double a, double b, double c) { return (a + b) + c; }

template <typename T> inline V3<T> sub(const V3<T>& a, const V3<T>& b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
template <typename T> inline V3<T> add(const V3<T>& a, const V3<T>& b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
template <typename T> inline V3<T> scale(const V3<T>& a, T s) { return {a.x * s, a.y * s, a.z * s}; }
template <typename T> inline V3<T> divs(const V3<T>& a, T s) { return {a.x / s, a.y / s, a.z / s}; }
template <typename T> inline T dot(const V3<T>& a, const V3<T>& b) { return sum3<T>(a.x * b.x, a.y * b.y, a.z * b.z); }
template <typename T> inline T sqnorm(const V3<T>& a) { return sum3<T>(a.x * a.x, a.y * a.y, a.z * a.z); }
template <typename T> inline T norm(const V3<T>& a) { return std::sqrt(sqnorm(a)); }
template <typename T> inline V3<T> cross(const V3<T>& a, const V3<T>& b)
{
  return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
// Eigen::MatrixBase::normalized(): returns the vector itself when its norm is 0
template <typename T> inline V3<T> normalized(const V3<T>& a)
{
  T z = sqnorm(a);
  if (z > T(0)) return divs(a, std::sqrt(z));
  return a;
}
inline V3f xyz(const Point& p) { return {p.x, p.y, p.z}; }

// ---------------------------------------------------------------------------
// 3x3 matrix, row-major m[r*3+c]
template <typename T> struct M3 { T m[9]; T& operator()(int r, int c) { return m[r * 3 + c]; } T operator()(int r, int c) const { return m[r * 3 + c]; } };
using M3d = M3<double>;

template <typename T> inline V3<T> row(const M3<T>& a, int r) { return {a(r, 0), a(r, 1), a(r, 2)}; }
template <typename T> inline V3<T> col(const M3<T>& a, int c) { return {a(0, c), a(1, c), a(2, c)}; }
template <typename T> inline void setcol(M3<T>& a, int c, const V3<T>& v) { a(0, c) = v.x; a(1, c) = v.y; a(2, c) = v.z; }

// ---------------------------------------------------------------------------
// pcl::computeMeanAndCovarianceMatrix restated (PCL 1.10, dense cloud branch):
// the coordinate products are float*float (PointT members are float) and are
// then added to a Scalar accumulator.  idx[] gives the point order.
template <typename T, typename GetXYZ>
inline void mean_and_cov(int n, GetXYZ get, V3<T>& centroid, M3<T>& cov)
{
  T accu[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  for (int i = 0; i < n; ++i)
  {
    float px, py, pz;
    get(i, px, py, pz);
    accu[0] += px * px;
    accu[1] += px * py;
    accu[2] += px * pz;
    accu[3] += py * py;
    accu[4] += py * pz;
    accu[5] += pz * pz;
    accu[6] += px;
    accu[7] += py;
    accu[8] += pz;
  }
  const T cnt = static_cast<T>(n);
  for (int i = 0; i < 9; ++i) accu[i] /= cnt;
  centroid = {accu[6], accu[7], accu[8]};
  cov(0, 0) = accu[0] - accu[6] * accu[6];
  cov(0, 1) = accu[1] - accu[6] * accu[7];
  cov(0, 2) = accu[2] - accu[6] * accu[8];
  cov(1, 1) = accu[3] - accu[7] * accu[7];
  cov(1, 2) = accu[4] - accu[7] * accu[8];
  cov(2, 2) = accu[5] - accu[8] * accu[8];
  cov(1, 0) = cov(0, 1);
  cov(2, 0) = cov(0, 2);
  cov(2, 1) = cov(1, 2);
}

// ---------------------------------------------------------------------------
// pcl::computeRoots2 / computeRoots / eigen33 restated (recalled from upstream
// pcl/common/impl/eigen.hpp -- not verifiable offline).
template <typename T> inline void compute_roots2(T b, T c, T roots[3])
{
  roots[0] = T(0);
  T d = T(b * b - 4.0 * c);
  if (d < 0.0) d = 0.0;
  T sd = std::sqrt(d);
  roots[2] = 0.5f * (b + sd);
  roots[1] = 0.5f * (b - sd);
}

template <typename T> inline void compute_roots(const M3<T>& m, T roots[3])
{
  T c0 = m(0, 0) * m(1, 1) * m(2, 2) + T(2) * m(0, 1) * m(0, 2) * m(1, 2) - m(0, 0) * m(1, 2) * m(1, 2) -
         m(1, 1) * m(0, 2) * m(0, 2) - m(2, 2) * m(0, 1) * m(0, 1);
  T c1 = m(0, 0) * m(1, 1) - m(0, 1) * m(0, 1) + m(0, 0) * m(2, 2) - m(0, 2) * m(0, 2) + m(1, 1) * m(2, 2) -
         m(1, 2) * m(1, 2);
  T c2 = m(0, 0) + m(1, 1) + m(2, 2);

  if (std::abs(c0) < std::numeric_limits<T>::epsilon())
  {
    compute_roots2(c2, c1, roots);
    return;
  }
  const T s_inv3 = T(1.0 / 3.0);
  const T s_sqrt3 = T(1.7320508075688772935);
  T c2_over_3 = c2 * s_inv3;
  T a_over_3 = (c1 - c2 * c2_over_3) * s_inv3;
  if (a_over_3 > T(0)) a_over_3 = T(0);
  T half_b = T(0.5) * (c0 + c2_over_3 * (T(2) * c2_over_3 * c2_over_3 - c1));
  T q = half_b * half_b + a_over_3 * a_over_3 * a_over_3;
  if (q > T(0)) q = T(0);

  T rho = std::sqrt(-a_over_3);
  T theta = T(t_atan2((double)std::sqrt(-q), (double)half_b)) * s_inv3;
  T cos_theta = T(t_cos((double)theta));
  T sin_theta = T(t_sin((double)theta));
  roots[0] = c2_over_3 + T(2) * rho * cos_theta;
  roots[1] = c2_over_3 - rho * (cos_theta + s_sqrt3 * sin_theta);
  roots[2] = c2_over_3 - rho * (cos_theta - s_sqrt3 * sin_theta);

  if (roots[0] >= roots[1]) std::swap(roots[0], roots[1]);
  if (roots[1] >= roots[2])
  {
    std::swap(roots[1], roots[2]);
    if (roots[0] >= roots[1]) std::swap(roots[0], roots[1]);
  }
  if (roots[0] <= 0) compute_roots2(c2, c1, roots);
}

// Eigen unitOrthogonal() for 3-vectors
template <typename T> inline V3<T> unit_orthogonal(const V3<T>& s)
{
  const T prec = std::is_same<T, float>::value ? T(1e-5) : T(1e-12);
  auto much_smaller = [prec](T a, T b) { return std::abs(a) <= std::abs(b) * prec; };
  V3<T> p;
  if (!much_smaller(s.x, s.z) || !much_smaller(s.y, s.z))
  {
    T invnm = T(1) / std::sqrt(s.x * s.x + s.y * s.y);
    p.x = -s.y * invnm;
    p.y = s.x * invnm;
    p.z = 0;
  }
  else
  {
    T invnm = T(1) / std::sqrt(s.y * s.y + s.z * s.z);
    p.x = 0;
    p.y = -s.z * invnm;
    p.z = s.y * invnm;
  }
  return p;
}

// helper of eigen33: largest of the three row cross products of (M - lambda I)
template <typename T> inline V3<T> best_null_vector(const M3<T>& scaled, T lambda, T& len)
{
  M3<T> tmp = scaled;
  tmp(0, 0) -= lambda;
  tmp(1, 1) -= lambda;
  tmp(2, 2) -= lambda;
  V3<T> v1 = cross(row(tmp, 0), row(tmp, 1));
  V3<T> v2 = cross(row(tmp, 0), row(tmp, 2));
  V3<T> v3 = cross(row(tmp, 1), row(tmp, 2));
  T l1 = sqnorm(v1), l2 = sqnorm(v2), l3 = sqnorm(v3);
  if (l1 >= l2 && l1 >= l3) { len = l1; return divs(v1, std::sqrt(l1)); }
  if (l2 >= l1 && l2 >= l3) { len = l2; return divs(v2, std::sqrt(l2)); }
  len = l3;
  return divs(v3, std::sqrt(l3));
}

// evecs columns = eigenvectors, evals ascending
template <typename T> inline void eigen33(const M3<T>& mat, M3<T>& evecs, T evals[3])
{
  T scale = 0;
  for (int i = 0; i < 9; ++i) scale = std::max(scale, std::abs(mat.m[i]));
  if (scale <= std::numeric_limits<T>::min()) scale = T(1.0);
  M3<T> sm;
  for (int i = 0; i < 9; ++i) sm.m[i] = mat.m[i] / scale;

  compute_roots(sm, evals);
  const T eps = std::numeric_limits<T>::epsilon();
  T len;
  if ((evals[2] - evals[0]) <= eps)
  {
    for (int i = 0; i < 9; ++i) evecs.m[i] = 0;
    evecs(0, 0) = evecs(1, 1) = evecs(2, 2) = 1;
  }
  else if ((evals[1] - evals[0]) <= eps)
  {
    V3<T> e2 = best_null_vector(sm, evals[2], len);
    V3<T> e1 = unit_orthogonal(e2);
    V3<T> e0 = cross(e1, e2);
    setcol(evecs, 2, e2); setcol(evecs, 1, e1); setcol(evecs, 0, e0);
  }
  else if ((evals[2] - evals[1]) <= eps)
  {
    V3<T> e0 = best_null_vector(sm, evals[0], len);
    V3<T> e1 = unit_orthogonal(e0);
    V3<T> e2 = cross(e0, e1);
    setcol(evecs, 0, e0); setcol(evecs, 1, e1); setcol(evecs, 2, e2);
  }
  else
  {
    T mmax[3];
    unsigned min_el = 2, max_el = 2;
    V3<T> e[3];
    e[2] = best_null_vector(sm, evals[2], len);
    mmax[2] = len;
    e[1] = best_null_vector(sm, evals[1], len);
    mmax[1] = len;
    min_el = len <= mmax[min_el] ? 1 : min_el;
    max_el = len > mmax[max_el] ? 1 : max_el;
    e[0] = best_null_vector(sm, evals[0], len);
    mmax[0] = len;
    min_el = len <= mmax[min_el] ? 0 : min_el;
    max_el = len > mmax[max_el] ? 0 : max_el;
    unsigned mid_el = 3 - min_el - max_el;
    e[min_el] = normalized(cross(e[(min_el + 1) % 3], e[(min_el + 2) % 3]));
    if (mid_el < 3)  // NaN input leaves min_el == max_el: the reference then indexes out of range
      e[mid_el] = normalized(cross(e[(mid_el + 1) % 3], e[(mid_el + 2) % 3]));
    setcol(evecs, 0, e[0]); setcol(evecs, 1, e[1]); setcol(evecs, 2, e[2]);
  }
  evals[0] *= scale; evals[1] *= scale; evals[2] *= scale;
}

// ---------------------------------------------------------------------------
// Rigid transforms (Eigen::Isometry3d): R row-major 3x3 + t
struct Iso
{
  double R[9];
  double t[3];
};
inline Iso iso_identity() { return {{1, 0, 0, 0, 1, 0, 0, 0, 1}, {0, 0, 0}}; }
// Eigen small fixed product: ((a0*b0 + a1*b1) + a2*b2)
inline Iso iso_mul(const Iso& a, const Iso& b)
{
  Iso r;
  for (int i = 0; i < 3; ++i)
  {
    for (int j = 0; j < 3; ++j)
      r.R[i * 3 + j] = (a.R[i * 3 + 0] * b.R[0 * 3 + j] + a.R[i * 3 + 1] * b.R[1 * 3 + j]) + a.R[i * 3 + 2] * b.R[2 * 3 + j];
    r.t[i] = ((a.R[i * 3 + 0] * b.t[0] + a.R[i * 3 + 1] * b.t[1]) + a.R[i * 3 + 2] * b.t[2]) + a.t[i];
  }
  return r;
}
inline Iso iso_inverse(const Iso& a)
{
  Iso r;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) r.R[i * 3 + j] = a.R[j * 3 + i];
  for (int i = 0; i < 3; ++i)
    r.t[i] = -((r.R[i * 3 + 0] * a.t[0] + r.R[i * 3 + 1] * a.t[1]) + r.R[i * 3 + 2] * a.t[2]);
  return r;
}
// T * [x y z 1]: ((R0 x + R1 y) + R2 z) + t
inline V3d iso_apply(const Iso& a, const V3d& p)
{
  return {((a.R[0] * p.x + a.R[1] * p.y) + a.R[2] * p.z) + a.t[0],
          ((a.R[3] * p.x + a.R[4] * p.y) + a.R[5] * p.z) + a.t[1],
          ((a.R[6] * p.x + a.R[7] * p.y) + a.R[8] * p.z) + a.t[2]};
}
// Utils::TransformPoint (slam_lib/include/LidarSlam/Utilities.h:274-278):
// double math, float store
inline void transform_point(Point& p, const Iso& a)
{
  V3d q = iso_apply(a, {(double)p.x, (double)p.y, (double)p.z});
  p.x = (float)q.x; p.y = (float)q.y; p.z = (float)q.z;
}
// Eigen isApprox on the 4x4 matrices (prec 1e-12)
inline bool iso_is_approx(const Iso& a, const Iso& b)
{
  double d = 0, na = 1, nb = 1;  // bottom-right 1 contributes to both norms
  for (int i = 0; i < 9; ++i) { double e = a.R[i] - b.R[i]; d += e * e; na += a.R[i] * a.R[i]; nb += b.R[i] * b.R[i]; }
  for (int i = 0; i < 3; ++i) { double e = a.t[i] - b.t[i]; d += e * e; na += a.t[i] * a.t[i]; nb += b.t[i] * b.t[i]; }
  return d <= 1e-12 * 1e-12 * std::min(na, nb);
}

// ---------------------------------------------------------------------------
// Quaternion (w, x, y, z)
struct Quat { double w, x, y, z; };

inline Quat quat_from_matrix(const double R[9])
{
  Quat q;
  double t = R[0] + R[4] + R[8];
  if (t > 0.0)
  {
    t = std::sqrt(t + 1.0);
    q.w = 0.5 * t;
    t = 0.5 / t;
    q.x = (R[7] - R[5]) * t;
    q.y = (R[2] - R[6]) * t;
    q.z = (R[3] - R[1]) * t;
  }
  else
  {
    int i = 0;
    if (R[4] > R[0]) i = 1;
    if (R[8] > R[i * 3 + i]) i = 2;
    int j = (i + 1) % 3, k = (j + 1) % 3;
    t = std::sqrt(R[i * 3 + i] - R[j * 3 + j] - R[k * 3 + k] + 1.0);
    double v[3];
    v[i] = 0.5 * t;
    t = 0.5 / t;
    q.w = (R[k * 3 + j] - R[j * 3 + k]) * t;
    v[j] = (R[j * 3 + i] + R[i * 3 + j]) * t;
    v[k] = (R[k * 3 + i] + R[i * 3 + k]) * t;
    q.x = v[0]; q.y = v[1]; q.z = v[2];
  }
  return q;
}
inline void quat_to_matrix(const Quat& q, double R[9])
{
  const double tx = 2.0 * q.x, ty = 2.0 * q.y, tz = 2.0 * q.z;
  const double twx = tx * q.w, twy = ty * q.w, twz = tz * q.w;
  const double txx = tx * q.x, txy = ty * q.x, txz = tz * q.x;
  const double tyy = ty * q.y, tyz = tz * q.y, tzz = tz * q.z;
  R[0] = 1.0 - (tyy + tzz); R[1] = txy - twz;         R[2] = txz + twy;
  R[3] = txy + twz;         R[4] = 1.0 - (txx + tzz); R[5] = tyz - twx;
  R[6] = txz - twy;         R[7] = tyz + twx;         R[8] = 1.0 - (txx + tyy);
}
inline Quat quat_mul(const Quat& a, const Quat& b)
{
  return {a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z,
          a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y,
          a.w * b.y + a.y * b.w + a.z * b.x - a.x * b.z,
          a.w * b.z + a.z * b.w + a.x * b.y - a.y * b.x};
}
// Constants of a slerp between two fixed quaternions (everything that does not
// depend on t), in the portable functions the GPU evaluates them with.
struct SlerpConst { Quat a, b; double d, theta, sin_theta; bool linear; };
inline SlerpConst slerp_prepare(const Quat& a, const Quat& b)
{
  SlerpConst c;
  c.a = a; c.b = b;
  c.d = (a.x * b.x + a.z * b.z) + (a.y * b.y + a.w * b.w);  // Eigen Vector4d dot: Packet2d lanes (x,y)+(z,w), then predux
  const double one = 1.0 - std::numeric_limits<double>::epsilon();
  double absD = std::abs(c.d);
  c.linear = absD >= one;
  c.theta = c.linear ? 0.0 : t_acos(absD);
  c.sin_theta = c.linear ? 1.0 : t_sin(c.theta);
  return c;
}
inline Quat slerp_eval(const SlerpConst& c, double t)
{
  double s0, s1;
  if (c.linear) { s0 = 1.0 - t; s1 = t; }
  else
  {
    s0 = t_sin((1.0 - t) * c.theta) / c.sin_theta;
    s1 = t_sin(t * c.theta) / c.sin_theta;
  }
  if (c.d < 0.0) s1 = -s1;
  return {s0 * c.a.w + s1 * c.b.w, s0 * c.a.x + s1 * c.b.x, s0 * c.a.y + s1 * c.b.y, s0 * c.a.z + s1 * c.b.z};
}

// ---------------------------------------------------------------------------
// RPY <-> matrix (slam_lib/src/Utilities.cxx:33-77)
inline void rpy_to_matrix(double roll, double pitch, double yaw, double R[9])
{
  Quat qz = {t_cos(yaw * 0.5), 0, 0, t_sin(yaw * 0.5)};
  Quat qy = {t_cos(pitch * 0.5), 0, t_sin(pitch * 0.5), 0};
  Quat qx = {t_cos(roll * 0.5), t_sin(roll * 0.5), 0, 0};
  quat_to_matrix(quat_mul(quat_mul(qz, qy), qx), R);
}
inline void matrix_to_rpy(const double R[9], double rpy[3])
{
  rpy[0] = t_atan2(R[7], R[8]);
  rpy[1] = -t_asin(R[6]);
  rpy[2] = t_atan2(R[3], R[0]);
}
inline Iso xyzrpy_to_iso(const double w[6])
{
  Iso r;
  rpy_to_matrix(w[3], w[4], w[5], r.R);
  r.t[0] = w[0]; r.t[1] = w[1]; r.t[2] = w[2];
  return r;
}
inline void iso_to_xyzrpy(const Iso& a, double w[6])
{
  w[0] = a.t[0]; w[1] = a.t[1]; w[2] = a.t[2];
  matrix_to_rpy(a.R, w + 3);
}

// LinearInterpolation (slam_lib/src/MotionModel.cxx:26-34): returns H1 when
// t0 == t1 or H0 ~ H1
inline Iso linear_interpolation(const Iso& H0, const Iso& H1, double t, double t0, double t1)
{
  if (t0 == t1 || iso_is_approx(H0, H1)) return H1;
  const double time = (t - t0) / (t1 - t0);
  SlerpConst sc = slerp_prepare(quat_from_matrix(H0.R), quat_from_matrix(H1.R));
  Iso r;
  quat_to_matrix(slerp_eval(sc, time), r.R);
  for (int i = 0; i < 3; ++i) r.t[i] = H0.t[i] + time * (H1.t[i] - H0.t[i]);
  return r;
}

// LinearTransformInterpolator<double> (slam_lib/include/LidarSlam/MotionModel.h:36-136):
// keeps the rotations as quaternions, returns H0 when invalid.
struct Interpolator
{
  double Time0 = 0., Time1 = 1.;
  Quat Rot0{1, 0, 0, 0}, Rot1{1, 0, 0, 0};
  double Trans0[3] = {0, 0, 0}, Trans1[3] = {0, 0, 0};
  bool IsInvalid = true;

  Iso GetH0() const { Iso r; quat_to_matrix(Rot0, r.R); std::memcpy(r.t, Trans0, sizeof(r.t)); return r; }
  Iso GetH1() const { Iso r; quat_to_matrix(Rot1, r.R); std::memcpy(r.t, Trans1, sizeof(r.t)); return r; }
  void Revalidate() { IsInvalid = (Time0 == Time1) || iso_is_approx(GetH0(), GetH1()); }
  void SetTransforms(const Iso& H0, const Iso& H1)
  {
    Rot0 = quat_from_matrix(H0.R); std::memcpy(Trans0, H0.t, sizeof(Trans0));
    Rot1 = quat_from_matrix(H1.R); std::memcpy(Trans1, H1.t, sizeof(Trans1));
    Revalidate();
  }
  void SetTimes(double t0, double t1) { Time0 = t0; Time1 = t1; Revalidate(); }
  double GetTimeRange() const { return Time1 - Time0; }
  Iso operator()(double t) const
  {
    if (IsInvalid) return GetH0();
    const double time = (t - Time0) / (Time1 - Time0);
    SlerpConst sc = slerp_prepare(Rot0, Rot1);
    Iso r;
    quat_to_matrix(slerp_eval(sc, time), r.R);
    for (int i = 0; i < 3; ++i) r.t[i] = Trans0[i] + time * (Trans1[i] - Trans0[i]);
    return r;
  }
};

}  // namespace orc
