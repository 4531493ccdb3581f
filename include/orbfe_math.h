/*
 * orbfe_math.h -- the numerics contract of the front-end.
 *
 * The few floating-point primitives whose results feed integer decisions
 * (descriptor bits, keypoint angles, rounded coordinates) are defined ONCE here
 * and compiled unchanged by g++ (host, oracle) and hipcc (gfx950 device code).
 * Every operation is a single IEEE-754 add/sub/mul/div on float or double, so
 * host and device agree bit for bit provided contraction is off
 * (-ffp-contract=off on both compilers; the Makefile and build() pass it).
 *
 * Reference call sites these restate:
 *   cvRound      -- src/ORBextractor.cc:81,117-120,442,1112 (OpenCV: round half to even)
 *   fastAtan2    -- src/ORBextractor.cc:103               (OpenCV 3.4 mathfuncs_core, scalar path)
 *   cos/sin      -- src/ORBextractor.cc:112-113           (libm cosf/sinf on a float radian angle)
 *
 * cos/sin: the reference calls libm, which is not reproducible across libm
 * builds nor available bit-identically on the GPU. orbfe_sincosf() evaluates
 * both in double (Cody-Waite reduction by pi/2 + the classic degree-13/14
 * minimax kernels, error < 1e-16) and rounds once to float. That equals the
 * correctly rounded float result except when the true value lies within
 * ~1e-16 relative of a float rounding boundary; glibc's cosf/sinf have the same
 * property, and tests/test_oracle_cpu.py checks agreement with the host libm
 * over a dense angle sweep.
 */
#ifndef ORBFE_MATH_H
#define ORBFE_MATH_H

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define ORBFE_HD __host__ __device__ inline
#else
#define ORBFE_HD inline
#endif

/* round half to even (the rounding mode is never changed from the IEEE default;
 * rint lowers to v_rndne_f64 on gfx950 and to libm/roundsd on the host). */
ORBFE_HD int orbfe_round_d(double v) { return (int)__builtin_rint(v); }
ORBFE_HD int orbfe_round_f(float v) { return (int)__builtin_rint((double)v); }
ORBFE_HD int orbfe_floor_d(double v) { return (int)__builtin_floor(v); }
ORBFE_HD int orbfe_ceil_d(double v) { return (int)__builtin_ceil(v); }

/* OpenCV fastAtan2 (degrees), scalar path. */
ORBFE_HD float orbfe_fast_atan2(float y, float x)
{
    const float scale = (float)(180.0 / 3.14159265358979323846);
    const float p1 = 0.9997878412794807f * scale;
    const float p3 = -0.3258083974640975f * scale;
    const float p5 = 0.1555786518463281f * scale;
    const float p7 = -0.04432655554792128f * scale;
    const float eps = (float)2.2204460492503131e-16; /* (float)DBL_EPSILON */
    float ax = x < 0 ? -x : x, ay = y < 0 ? -y : y;
    float a, c, c2;
    if (ax >= ay) {
        c = ay / (ax + eps);
        c2 = c * c;
        a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    } else {
        c = ax / (ay + eps);
        c2 = c * c;
        a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    }
    if (x < 0) a = 180.f - a;
    if (y < 0) a = 360.f - a;
    return a;
}

/* sin and cos of a float angle in radians, |x| <= ~1e4, each rounded once. */
ORBFE_HD void orbfe_sincosf(float xf, float* s, float* c)
{
    const double x = (double)xf;
    const double two_over_pi = 6.36619772367581382433e-01;
    const double pio2_hi = 1.57079632673412561417e+00; /* first 33 bits of pi/2 */
    const double pio2_lo = 6.07710050650619224932e-11; /* pi/2 - pio2_hi */
    const double pio2_lo2 = 2.02226624879595063154e-21;
    int n = orbfe_round_d(x * two_over_pi);
    double fn = (double)n;
    double r = (x - fn * pio2_hi) - fn * pio2_lo;
    r = r - fn * pio2_lo2;
    double z = r * r;
    /* sin kernel */
    const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03,
                 S3 = -1.98412698298579493134e-04, S4 = 2.75573137070700676789e-06,
                 S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
    double ps = S1 + z * (S2 + z * (S3 + z * (S4 + z * (S5 + z * S6))));
    double sn = r + r * z * ps;
    /* cos kernel */
    const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03,
                 C3 = 2.48015872894767294178e-05, C4 = -2.75573143513906633035e-07,
                 C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
    double pc = z * (C1 + z * (C2 + z * (C3 + z * (C4 + z * (C5 + z * C6)))));
    double cs = (1.0 - 0.5 * z) + z * pc;
    double rs, rc;
    switch (n & 3) {
    case 0: rs = sn; rc = cs; break;
    case 1: rs = cs; rc = -sn; break;
    case 2: rs = -sn; rc = -cs; break;
    default: rs = -cs; rc = sn; break;
    }
    *s = (float)rs;
    *c = (float)rc;
}


/* cv::undistortPoints(src, dst, K, distCoeffs, noArray(), K) of OpenCV 3.4 for one point, as Frame::UndistortKeyPoints
 * (Frame.cc:357-387), UndistortArucoCorners (:389-416) and ComputeImageBounds (:418-451) call it: normalise with K,
 * five fixed-point iterations of the inverse Brown model (radial k1 k2 k3 [k4 k5 k6], tangential p1 p2, thin prism
 * s1..s4 -- ORB_SLAM2 passes 4 or 5 coefficients, the rest are 0), re-project with P = K.  All arithmetic in double, the
 * result rounded to float once per coordinate.  k[] = the 12 coefficients in OpenCV's order k1 k2 p1 p2 k3 k4 k5 k6 s1..s4. */
ORBFE_HD void orbfe_undistort_point(float u, float v, double fx, double fy, double cx, double cy, const double k[12], float* ou,
                                    float* ov)
{
    const double ifx = 1. / fx, ify = 1. / fy;
    double x = ((double)u - cx) * ifx, y = ((double)v - cy) * ify;
    const double x0 = x, y0 = y;
    for (int j = 0; j < 5; j++) {
        const double r2 = x * x + y * y;
        const double icdist = (1 + ((k[7] * r2 + k[6]) * r2 + k[5]) * r2) / (1 + ((k[4] * r2 + k[1]) * r2 + k[0]) * r2);
        const double deltaX = 2 * k[2] * x * y + k[3] * (r2 + 2 * x * x) + k[8] * r2 + k[9] * r2 * r2;
        const double deltaY = k[2] * (r2 + 2 * y * y) + 2 * k[3] * x * y + k[10] * r2 + k[11] * r2 * r2;
        x = (x0 - deltaX) * icdist;
        y = (y0 - deltaY) * icdist;
    }
    const double xx = fx * x + 0. * y + cx, yy = 0. * x + fy * y + cy, ww = 1. / (0. * x + 0. * y + 1.);
    *ou = (float)(xx * ww);
    *ov = (float)(yy * ww);
}

#endif /* ORBFE_MATH_H */
