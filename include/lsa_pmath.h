/*
 * lsa_pmath.h -- portable elementary functions (double precision sin / cos /
 * atan / atan2) built ONLY from IEEE-754 correctly rounded operations
 * (+ - * / and comparisons).
 *
 * Why this exists (SURVEY.md hard part H1): the reference's analytic 3x3
 * eigen-solver (pcl::eigen33 -> computeRoots, reached from
 * slam_lib/include/LidarSlam/Utilities.h:257-261) and its per-point slerp
 * (slam_lib/include/LidarSlam/MotionModel.h:115-129) call std::atan2 /
 * std::cos / std::sin.  glibc and the ROCm device library differ in the last
 * ulp, so a decision next to a threshold could flip between the CPU oracle and
 * the GPU.  Every routine that is evaluated on the DEVICE therefore uses these
 * functions, and the CPU oracle uses the very same ones, compiled with
 * -ffp-contract=off on both sides: results are bit-identical by construction.
 * float callers evaluate in double and round once to float.
 *
 * Polynomial kernels and break points follow the classic public-domain
 * fdlibm constants (k_sin.c / k_cos.c / s_atan.c); accuracy < 1 ulp, checked
 * against libm in tests/test_pmath.py.
 *
 * This header is a numerical primitive shared by product and oracle; it holds
 * no part of the reference's algorithm.
 */
#ifndef LSA_PMATH_H
#define LSA_PMATH_H

#if defined(__HIPCC__)
#define LSA_HD __host__ __device__ inline
#else
#define LSA_HD static inline
#endif

/* |x| <= pi/4 */
LSA_HD double lsa_ksin(double x)
{
  const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03,
               S3 = -1.98412698298579493134e-04, S4 = 2.75573137070700676789e-06,
               S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
  double z = x * x;
  double v = z * x;
  double r = S2 + z * (S3 + z * (S4 + z * (S5 + z * S6)));
  return x + v * (S1 + z * r);
}

/* |x| <= pi/4 */
LSA_HD double lsa_kcos(double x)
{
  const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03,
               C3 = 2.48015872894767294178e-05, C4 = -2.75573143513906633035e-07,
               C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
  double z = x * x;
  double r = z * (C1 + z * (C2 + z * (C3 + z * (C4 + z * (C5 + z * C6)))));
  double hz = 0.5 * z;
  double w = 1.0 - hz;
  return w + (((1.0 - w) - hz) + z * r);
}

/* Cody-Waite reduction to [-pi/4, pi/4]; valid for |x| < ~1e5 (far beyond the
 * angles met on this path: [0, pi/3] in the eigen-solver, [0, pi] in slerp). */
LSA_HD double lsa_reduce_pio2(double x, int* quadrant)
{
  const double INV_PIO2 = 6.36619772367581382433e-01;
  const double PIO2_1 = 1.57079632673412561417e+00;  /* first 33 bits of pi/2 */
  const double PIO2_1T = 6.07710050650619224932e-11; /* pi/2 - PIO2_1 */
  double fn = x * INV_PIO2;
  /* round to nearest integer without libm */
  fn = (fn >= 0.0) ? (double)(long long)(fn + 0.5) : -(double)(long long)(0.5 - fn);
  *quadrant = (int)((long long)fn & 3LL);
  return (x - fn * PIO2_1) - fn * PIO2_1T;
}

LSA_HD double lsa_sin(double x)
{
  if (x != x) return x;
  if (x > -0.7853981633974483 && x < 0.7853981633974483) return lsa_ksin(x);
  int q;
  double r = lsa_reduce_pio2(x, &q);
  switch (q)
  {
    case 0: return lsa_ksin(r);
    case 1: return lsa_kcos(r);
    case 2: return -lsa_ksin(r);
    default: return -lsa_kcos(r);
  }
}

LSA_HD double lsa_cos(double x)
{
  if (x != x) return x;
  if (x > -0.7853981633974483 && x < 0.7853981633974483) return lsa_kcos(x);
  int q;
  double r = lsa_reduce_pio2(x, &q);
  switch (q)
  {
    case 0: return lsa_kcos(r);
    case 1: return -lsa_ksin(r);
    case 2: return -lsa_kcos(r);
    default: return lsa_ksin(r);
  }
}

/* atan for x >= 0 (finite or +inf) */
LSA_HD double lsa_atan_pos(double x)
{
  const double aT0 = 3.33333333333329318027e-01, aT1 = -1.99999999998764832476e-01,
               aT2 = 1.42857142725034663711e-01, aT3 = -1.11111104054623557880e-01,
               aT4 = 9.09088713343650656196e-02, aT5 = -7.69187620504482999495e-02,
               aT6 = 6.66107313738753120669e-02, aT7 = -5.83357013379057348645e-02,
               aT8 = 4.97687799461593236017e-02, aT9 = -3.65315727442169155270e-02,
               aT10 = 1.62858201153657823623e-02;
  double hi, lo;
  int id;
  if (x >= 7.378697629483821e19) /* 2^66 (also +inf) */
    return 1.57079632679489655800e+00 + 6.12323399573676603587e-17;
  if (x < 0.4375)
  {
    if (x < 1.862645149230957e-09) return x; /* 2^-29 */
    id = -1; hi = 0.0; lo = 0.0;
  }
  else if (x < 1.1875)
  {
    if (x < 0.6875) { id = 0; hi = 4.63647609000806093515e-01; lo = 2.26987774529616870924e-17; x = (2.0 * x - 1.0) / (2.0 + x); }
    else            { id = 1; hi = 7.85398163397448278999e-01; lo = 3.06161699786838301793e-17; x = (x - 1.0) / (x + 1.0); }
  }
  else
  {
    if (x < 2.4375) { id = 2; hi = 9.82793723247329054082e-01; lo = 1.39033110312309984516e-17; x = (x - 1.5) / (1.0 + 1.5 * x); }
    else            { id = 3; hi = 1.57079632679489655800e+00; lo = 6.12323399573676603587e-17; x = -1.0 / x; }
  }
  double z = x * x;
  double w = z * z;
  double s1 = z * (aT0 + w * (aT2 + w * (aT4 + w * (aT6 + w * (aT8 + w * aT10)))));
  double s2 = w * (aT1 + w * (aT3 + w * (aT5 + w * (aT7 + w * aT9))));
  if (id < 0) return x - x * (s1 + s2);
  return hi - ((x * (s1 + s2) - lo) - x);
}

LSA_HD double lsa_atan2(double y, double x)
{
  const double PI = 3.1415926535897931160e+00, PI_LO = 1.2246467991473531772e-16;
  const double PIO2 = 1.57079632679489655800e+00;
  if (x != x || y != y) return x + y;
  double ay = y < 0.0 ? -y : y;
  double ax = x < 0.0 ? -x : x;
  double z;
  if (ay == 0.0)
    z = 0.0;
  else if (ax == 0.0)
    z = PIO2 + 0.5 * PI_LO;
  else
  {
    const double INF = 1.0 / 0.0;
    if (ax == INF && ay == INF) z = 7.85398163397448278999e-01;
    else if (ax == INF) z = 0.0;
    else z = lsa_atan_pos(ay / ax);
  }
  /* x negative (including -0 when y == 0 is NOT distinguished: callers on this
   * path never pass signed zeros where it matters) */
  if (x < 0.0) z = PI - (z - PI_LO);
  return (y < 0.0) ? -z : z;
}

#endif /* LSA_PMATH_H */
