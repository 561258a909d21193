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

/* asin / acos: the rational approximation R(x^2) of fdlibm's e_asin.c / e_acos.c.  They serve the pose algebra
 * between two ICP iterations (roll-pitch-yaw of a rotation matrix, the angle of a slerp), which the device evaluates
 * behind a solve (lsa_lm.hip) and the host and the oracle evaluate with the same routine.  sqrt is the IEEE one on
 * both sides (__builtin_sqrt). */
LSA_HD double lsa_asin_r(double t)
{
  const double pS0 = 1.66666666666666657415e-01, pS1 = -3.25565818622400915405e-01, pS2 = 2.01212532134862925881e-01,
               pS3 = -4.00555345006794114027e-02, pS4 = 7.91534994289814532176e-04, pS5 = 3.47933107596021167570e-05,
               qS1 = -2.40339491173441421878e+00, qS2 = 2.02094576023350569471e+00, qS3 = -6.88283971605453293030e-01,
               qS4 = 7.70381505559019352791e-02;
  const double p = t * (pS0 + t * (pS1 + t * (pS2 + t * (pS3 + t * (pS4 + t * pS5)))));
  const double q = 1.0 + t * (qS1 + t * (qS2 + t * (qS3 + t * qS4)));
  return p / q;
}
/* v with the low 32 bits of its mantissa cleared */
LSA_HD double lsa_high_part(double v)
{
  unsigned long long b;
  __builtin_memcpy(&b, &v, sizeof(b));
  b &= 0xffffffff00000000ULL;
  __builtin_memcpy(&v, &b, sizeof(b));
  return v;
}

LSA_HD double lsa_asin(double x)
{
  const double PIO2_HI = 1.57079632679489655800e+00, PIO2_LO = 6.12323399573676603587e-17, PIO4_HI = 7.85398163397448278999e-01;
  if (x != x) return x;
  const double ax = x < 0.0 ? -x : x;
  if (ax >= 1.0)
  {
    if (ax == 1.0) return x * PIO2_HI + x * PIO2_LO;
    return (x - x) / (x - x); /* NaN */
  }
  if (ax < 0.5)
  {
    if (ax < 7.450580596923828125e-09) return x; /* 2^-27 */
    return x + x * lsa_asin_r(x * x);
  }
  const double w = 1.0 - ax;
  const double t = w * 0.5;
  const double r = lsa_asin_r(t);
  const double s = __builtin_sqrt(t);
  double res;
  if (ax >= 0.974999904632568359375) /* high word 0x3FEF3333 */
    res = PIO2_HI - (2.0 * (s + s * r) - PIO2_LO);
  else
  {
    const double sh = lsa_high_part(s);
    const double c = (t - sh * sh) / (s + sh);
    const double p = 2.0 * s * r - (PIO2_LO - 2.0 * c);
    const double q = PIO4_HI - 2.0 * sh;
    res = PIO4_HI - (p - q);
  }
  return x > 0.0 ? res : -res;
}

LSA_HD double lsa_acos(double x)
{
  const double PI = 3.14159265358979311600e+00, PIO2_HI = 1.57079632679489655800e+00, PIO2_LO = 6.12323399573676603587e-17;
  if (x != x) return x;
  const double ax = x < 0.0 ? -x : x;
  if (ax >= 1.0)
  {
    if (ax == 1.0) return x > 0.0 ? 0.0 : PI + 2.0 * PIO2_LO;
    return (x - x) / (x - x); /* NaN */
  }
  if (ax < 0.5)
  {
    if (ax <= 6.938893903907228377647697925567626953125e-18) return PIO2_HI + PIO2_LO; /* 2^-57 */
    const double r = lsa_asin_r(x * x);
    return PIO2_HI - (x - (PIO2_LO - r * x));
  }
  if (x < 0.0)
  {
    const double z = (1.0 + x) * 0.5;
    const double r = lsa_asin_r(z);
    const double s = __builtin_sqrt(z);
    const double w = r * s - PIO2_LO;
    return PI - 2.0 * (s + w);
  }
  const double z = (1.0 - x) * 0.5;
  const double s = __builtin_sqrt(z);
  const double df = lsa_high_part(s);
  const double c = (z - df * df) / (s + df);
  const double r = lsa_asin_r(z);
  const double w = r * s + c;
  return 2.0 * (df + w);
}

#endif /* LSA_PMATH_H */
