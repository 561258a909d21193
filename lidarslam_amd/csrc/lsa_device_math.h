// lsa_device_math.h -- fixed-size math used inside the HIP kernels.
//
// Every routine here is a restatement, in kernel-friendly form (registers only,
// no containers, no libm), of arithmetic the reference reaches through
// PCL / Eigen on its hot path.  Operation ORDER is part of the contract: the
// parity tests compare float decisions bit for bit against the CPU oracle, so
// nothing here may be re-associated and the translation unit is compiled with
// -ffp-contract=off.
//
//   pca3<T>()            pcl::computeMeanAndCovarianceMatrix + pcl::eigen33
//                        (slam_lib/include/LidarSlam/Utilities.h:247-262)
//   Eigen reductions     float: x + (y + z), double: (x + y) + z
//   normalized3()        Eigen::MatrixBase::normalized() (guards z > 0)
//   quat / slerp         Eigen::Quaternion::slerp, toRotationMatrix
//                        (slam_lib/include/LidarSlam/MotionModel.h:115-129)
// Transcendentals come from include/lsa_pmath.h (bit-identical on host/device).
#pragma once
#include <hip/hip_runtime.h>
#include "../../include/lsa_pmath.h"
#include "lsa_posemath.h"

#define LSA_DEV __device__ __forceinline__

namespace lsa
{

template <typename T> struct Limits;
template <> struct Limits<float>
{
  static LSA_DEV float eps() { return 1.1920928955078125e-07f; }
  static LSA_DEV float tiny() { return 1.17549435082228750797e-38f; }
  static LSA_DEV float ortho_prec() { return 1e-5f; }
};
template <> struct Limits<double>
{
  static LSA_DEV double eps() { return 2.220446049250313080847e-16; }
  static LSA_DEV double tiny() { return 2.2250738585072013830902e-308; }
  static LSA_DEV double ortho_prec() { return 1e-12; }
};

// llvm.sqrt: IEEE correctly rounded under hipcc's default -fhip-fp32-correctly-rounded-divide-sqrt
LSA_DEV float sqrt_t(float v) { return __builtin_sqrtf(v); }
LSA_DEV double sqrt_t(double v) { return __builtin_sqrt(v); }
LSA_DEV float abs_t(float v) { return fabsf(v); }
LSA_DEV double abs_t(double v) { return fabs(v); }

LSA_DEV float sum3(float a, float b, float c) { return a + (b + c); }
LSA_DEV double sum3(double a, double b, double c) { return (a + b) + c; }

template <typename T> struct Vec3
{
  T x, y, z;
};
template <typename T> LSA_DEV Vec3<T> vsub(const Vec3<T>& a, const Vec3<T>& b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
template <typename T> LSA_DEV T vsqnorm(const Vec3<T>& a) { return sum3(a.x * a.x, a.y * a.y, a.z * a.z); }
template <typename T> LSA_DEV T vnorm(const Vec3<T>& a) { return sqrt_t(vsqnorm(a)); }
template <typename T> LSA_DEV Vec3<T> vcross(const Vec3<T>& a, const Vec3<T>& b)
{
  return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
template <typename T> LSA_DEV Vec3<T> vdiv(const Vec3<T>& a, T s) { return {a.x / s, a.y / s, a.z / s}; }
template <typename T> LSA_DEV Vec3<T> normalized3(const Vec3<T>& a)
{
  T z = vsqnorm(a);
  if (z > T(0)) return vdiv(a, sqrt_t(z));
  return a;
}

// Symmetric 3x3 in 6 scalars + helpers to read rows of (M - lambda I)
template <typename T> struct Sym3
{
  T xx, xy, xz, yy, yz, zz;
};

template <typename T> LSA_DEV void roots2(T b, T c, T& r0, T& r1, T& r2)
{
  r0 = T(0);
  T d = T(b * b - 4.0 * c);
  if (d < 0.0) d = 0.0;
  T sd = sqrt_t(d);
  r2 = 0.5f * (b + sd);
  r1 = 0.5f * (b - sd);
}

template <typename T> LSA_DEV void swap_t(T& a, T& b) { T t = a; a = b; b = t; }

// pcl::computeRoots on the scaled matrix; roots ascending
template <typename T> LSA_DEV void roots3(const Sym3<T>& m, T& r0, T& r1, T& r2)
{
  T c0 = m.xx * m.yy * m.zz + T(2) * m.xy * m.xz * m.yz - m.xx * m.yz * m.yz - m.yy * m.xz * m.xz - m.zz * m.xy * m.xy;
  T c1 = m.xx * m.yy - m.xy * m.xy + m.xx * m.zz - m.xz * m.xz + m.yy * m.zz - m.yz * m.yz;
  T c2 = m.xx + m.yy + m.zz;
  if (abs_t(c0) < Limits<T>::eps())
  {
    roots2(c2, c1, r0, r1, r2);
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
  T rho = sqrt_t(-a_over_3);
  T theta = T(lsa_atan2((double)sqrt_t(-q), (double)half_b)) * s_inv3;
  T cos_theta = T(lsa_cos((double)theta));
  T sin_theta = T(lsa_sin((double)theta));
  r0 = c2_over_3 + T(2) * rho * cos_theta;
  r1 = c2_over_3 - rho * (cos_theta + s_sqrt3 * sin_theta);
  r2 = c2_over_3 - rho * (cos_theta - s_sqrt3 * sin_theta);
  if (r0 >= r1) swap_t(r0, r1);
  if (r1 >= r2)
  {
    swap_t(r1, r2);
    if (r0 >= r1) swap_t(r0, r1);
  }
  if (r0 <= 0) roots2(c2, c1, r0, r1, r2);
}

// largest-norm cross product of two rows of (M - lambda I), normalised
template <typename T> LSA_DEV Vec3<T> null_vector(const Sym3<T>& m, T lambda, T& len)
{
  Vec3<T> r0 = {m.xx - lambda, m.xy, m.xz};
  Vec3<T> r1 = {m.xy, m.yy - lambda, m.yz};
  Vec3<T> r2 = {m.xz, m.yz, m.zz - lambda};
  Vec3<T> v1 = vcross(r0, r1), v2 = vcross(r0, r2), v3 = vcross(r1, r2);
  T l1 = vsqnorm(v1), l2 = vsqnorm(v2), l3 = vsqnorm(v3);
  if (l1 >= l2 && l1 >= l3) { len = l1; return vdiv(v1, sqrt_t(l1)); }
  if (l2 >= l1 && l2 >= l3) { len = l2; return vdiv(v2, sqrt_t(l2)); }
  len = l3;
  return vdiv(v3, sqrt_t(l3));
}

template <typename T> LSA_DEV Vec3<T> unit_orthogonal(const Vec3<T>& s)
{
  const T prec = Limits<T>::ortho_prec();
  Vec3<T> p;
  if (!(abs_t(s.x) <= abs_t(s.z) * prec) || !(abs_t(s.y) <= abs_t(s.z) * prec))
  {
    T invnm = T(1) / sqrt_t(s.x * s.x + s.y * s.y);
    p.x = -s.y * invnm; p.y = s.x * invnm; p.z = 0;
  }
  else
  {
    T invnm = T(1) / sqrt_t(s.y * s.y + s.z * s.z);
    p.x = 0; p.y = -s.z * invnm; p.z = s.y * invnm;
  }
  return p;
}

// pcl::eigen33: e0/e1/e2 eigenvectors of ascending eigenvalues l0 <= l1 <= l2
template <typename T>
LSA_DEV void eigen33(const Sym3<T>& mat, Vec3<T>& e0, Vec3<T>& e1, Vec3<T>& e2, T& l0, T& l1, T& l2)
{
  T scale = abs_t(mat.xx);
  T a;
  a = abs_t(mat.xy); if (scale < a) scale = a;
  a = abs_t(mat.xz); if (scale < a) scale = a;
  a = abs_t(mat.yy); if (scale < a) scale = a;
  a = abs_t(mat.yz); if (scale < a) scale = a;
  a = abs_t(mat.zz); if (scale < a) scale = a;
  if (scale <= Limits<T>::tiny()) scale = T(1.0);
  Sym3<T> m = {mat.xx / scale, mat.xy / scale, mat.xz / scale, mat.yy / scale, mat.yz / scale, mat.zz / scale};
  roots3(m, l0, l1, l2);
  const T eps = Limits<T>::eps();
  T len;
  if ((l2 - l0) <= eps)
  {
    e0 = {1, 0, 0}; e1 = {0, 1, 0}; e2 = {0, 0, 1};
  }
  else if ((l1 - l0) <= eps)
  {
    e2 = null_vector(m, l2, len);
    e1 = unit_orthogonal(e2);
    e0 = vcross(e1, e2);
  }
  else if ((l2 - l1) <= eps)
  {
    e0 = null_vector(m, l0, len);
    e1 = unit_orthogonal(e0);
    e2 = vcross(e0, e1);
  }
  else
  {
    T m2, m1, m0;
    e2 = null_vector(m, l2, m2);
    e1 = null_vector(m, l1, m1);
    e0 = null_vector(m, l0, m0);
    // min/max bookkeeping of the reference, unrolled (indices 2 -> 1 -> 0)
    int min_el = 2, max_el = 2;
    T mmin = m2, mmax = m2;
    if (m1 <= mmin) { min_el = 1; mmin = m1; }
    if (m1 > mmax) { max_el = 1; mmax = m1; }
    if (m0 <= mmin) { min_el = 0; }
    if (m0 > mmax) { max_el = 0; }
    int mid_el = 3 - min_el - max_el;
    // e[min] = normalized(e[min+1] x e[min+2]); then e[mid] likewise with the updated vectors
    if (min_el == 0) e0 = normalized3(vcross(e1, e2));
    else if (min_el == 1) e1 = normalized3(vcross(e2, e0));
    else e2 = normalized3(vcross(e0, e1));
    if (mid_el == 0) e0 = normalized3(vcross(e1, e2));
    else if (mid_el == 1) e1 = normalized3(vcross(e2, e0));
    else if (mid_el == 2) e2 = normalized3(vcross(e0, e1));
  }
  l0 *= scale; l1 *= scale; l2 *= scale;
}

// Running sums of pcl::computeMeanAndCovarianceMatrix (PCL 1.10 dense branch:
// float products, Scalar accumulators)
template <typename T> struct CovAccum
{
  T a0 = 0, a1 = 0, a2 = 0, a3 = 0, a4 = 0, a5 = 0, a6 = 0, a7 = 0, a8 = 0;
  LSA_DEV void add(float x, float y, float z)
  {
    a0 += x * x; a1 += x * y; a2 += x * z; a3 += y * y; a4 += y * z; a5 += z * z;
    a6 += x; a7 += y; a8 += z;
  }
  LSA_DEV void finish(int n, Vec3<T>& mean, Sym3<T>& cov)
  {
    const T c = static_cast<T>(n);
    a0 /= c; a1 /= c; a2 /= c; a3 /= c; a4 /= c; a5 /= c; a6 /= c; a7 /= c; a8 /= c;
    mean = {a6, a7, a8};
    cov.xx = a0 - a6 * a6; cov.xy = a1 - a6 * a7; cov.xz = a2 - a6 * a8;
    cov.yy = a3 - a7 * a7; cov.yz = a4 - a7 * a8; cov.zz = a5 - a8 * a8;
  }
};

// (Rigid, InterpConst, IcpGate: lsa_posemath.h)
LSA_DEV void rigid_apply(const Rigid& a, double x, double y, double z, double& ox, double& oy, double& oz)
{
  ox = ((a.R[0] * x + a.R[1] * y) + a.R[2] * z) + a.t[0];
  oy = ((a.R[3] * x + a.R[4] * y) + a.R[5] * z) + a.t[1];
  oz = ((a.R[6] * x + a.R[7] * y) + a.R[8] * z) + a.t[2];
}

LSA_DEV void interp_eval(const InterpConst& c, double t, Rigid& out)
{
  if (c.invalid)
  {
    out = c.h0;
    return;
  }
  const double time = (t - c.time0) / (c.time1 - c.time0);
  double s0, s1;
  if (c.linear) { s0 = 1.0 - time; s1 = time; }
  else
  {
    s0 = lsa_sin((1.0 - time) * c.theta) / c.sin_theta;
    s1 = lsa_sin(time * c.theta) / c.sin_theta;
  }
  if (c.d < 0.0) s1 = -s1;
  const double w = s0 * c.qa[0] + s1 * c.qb[0];
  const double x = s0 * c.qa[1] + s1 * c.qb[1];
  const double y = s0 * c.qa[2] + s1 * c.qb[2];
  const double z = s0 * c.qa[3] + s1 * c.qb[3];
  const double tx = 2.0 * x, ty = 2.0 * y, tz = 2.0 * z;
  const double twx = tx * w, twy = ty * w, twz = tz * w;
  const double txx = tx * x, txy = ty * x, txz = tz * x;
  const double tyy = ty * y, tyz = tz * y, tzz = tz * z;
  out.R[0] = 1.0 - (tyy + tzz); out.R[1] = txy - twz;         out.R[2] = txz + twy;
  out.R[3] = txy + twz;         out.R[4] = 1.0 - (txx + tzz); out.R[5] = tyz - twx;
  out.R[6] = txz - twy;         out.R[7] = tyz + twx;         out.R[8] = 1.0 - (txx + tyy);
  out.t[0] = c.trans0[0] + time * (c.trans1[0] - c.trans0[0]);
  out.t[1] = c.trans0[1] + time * (c.trans1[1] - c.trans0[1]);
  out.t[2] = c.trans0[2] + time * (c.trans1[2] - c.trans0[2]);
}

}  // namespace lsa
