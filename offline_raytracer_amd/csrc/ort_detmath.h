/*
 * offline_raytracer_amd/csrc/ort_detmath.h -- deterministic sinf/cosf/atan2f/powf/logf for the
 * render path, host and gfx950 device.
 *
 * The reference calls libm on its hot path (code/ray.cpp:829,857,964-966,1072,1128,1138,
 * 1233-1234; code/random.h:107-110) and in scene ingestion (code/math.h:754-757,
 * code/parser.cpp:247).  libm bits differ between platforms and a GPU cannot reproduce
 * them, so this path computes those five functions from IEEE-754 binary64 + - * / and
 * integer operations only -- no fma, no sqrt, no tables -- which gives identical bits on
 * x86-64 and gfx950 when built with -ffp-contract=off.  Accuracy: ~1e-13 relative before
 * the final round to f32 (correctly rounded except in ~1e-6 of cases; within 1 ulp of
 * glibc 2.35 on the path's argument ranges).
 */
#ifndef ORT_DETMATH_H
#define ORT_DETMATH_H

#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define ORT_HD __host__ __device__
#else
#define ORT_HD
#endif

ORT_HD inline uint64_t om_f64_bits(double d) { return __builtin_bit_cast(uint64_t, d); }
ORT_HD inline double om_bits_f64(uint64_t u) { return __builtin_bit_cast(double, u); }
ORT_HD inline uint32_t om_f32_bits(float f) { return __builtin_bit_cast(uint32_t, f); }
ORT_HD inline float om_bits_f32(uint32_t u) { return __builtin_bit_cast(float, u); }

ORT_HD inline int om_isnan_f(float x) { return (om_f32_bits(x) & 0x7fffffffu) > 0x7f800000u; }
ORT_HD inline int om_isinf_f(float x) { return (om_f32_bits(x) & 0x7fffffffu) == 0x7f800000u; }
ORT_HD inline float om_nan_f(void) { return om_bits_f32(0x7fc00000u); }
ORT_HD inline float om_inf_f(void) { return om_bits_f32(0x7f800000u); }

#define OM_PIO2_HI 1.57079632673412561417e+00 /* first 33 bits of pi/2 */
#define OM_PIO2_LO 6.07710050650619224932e-11 /* pi/2 - OM_PIO2_HI */
#define OM_INV_PIO2 6.36619772367581382433e-01
#define OM_PI 3.14159265358979311600e+00
#define OM_PI_2 1.57079632679489655800e+00
#define OM_PI_4 7.85398163397448278999e-01
#define OM_LN2_HI 6.93147180369123816490e-01
#define OM_LN2_LO 1.90821492927058770002e-10
#define OM_INV_LN2 1.44269504088896338700e+00

/* sin(r), cos(r) for |r| <= pi/4 (+ a little); minimax coefficients as published
 * in FreeBSD msun k_sin.c / k_cos.c. */
ORT_HD inline double om_sin_k(double r)
{
    double z = r * r;
    double p = 1.58969099521155010221e-10;
    p = -2.50507602534068634195e-08 + z * p;
    p = 2.75573137070700676789e-06 + z * p;
    p = -1.98412698298579493134e-04 + z * p;
    p = 8.33333333332248946124e-03 + z * p;
    p = -1.66666666666666324348e-01 + z * p;
    return r + r * (z * p);
}

ORT_HD inline double om_cos_k(double r)
{
    double z = r * r;
    double p = -1.13596475577881948265e-11;
    p = 2.08757232129817482790e-09 + z * p;
    p = -2.75573143513906633035e-07 + z * p;
    p = 2.48015872894767294178e-05 + z * p;
    p = -1.38888888888741095749e-03 + z * p;
    p = 4.16666666666666019037e-02 + z * p;
    return (1.0 - 0.5 * z) + z * (z * p);
}

/* n = nearest integer to x*2/pi, *r = x - n*pi/2.  Valid for |x| < 1e9. */
ORT_HD inline int om_rem_pio2(double x, double *r)
{
    double fn = x * OM_INV_PIO2;
    int n = (int)(fn + (fn < 0.0 ? -0.5 : 0.5));
    double dn = (double)n;
    *r = (x - dn * OM_PIO2_HI) - dn * OM_PIO2_LO;
    return n;
}

ORT_HD inline float ort_sinf(float xf)
{
    double x = (double)xf, r, v;
    int n;
    if (!(x > -1.0e9 && x < 1.0e9)) return xf - xf; /* NaN for NaN/Inf, 0 for huge */
    n = om_rem_pio2(x, &r);
    switch (n & 3) {
    case 0: v = om_sin_k(r); break;
    case 1: v = om_cos_k(r); break;
    case 2: v = -om_sin_k(r); break;
    default: v = -om_cos_k(r); break;
    }
    return (float)v;
}

ORT_HD inline float ort_cosf(float xf)
{
    double x = (double)xf, r, v;
    int n;
    if (!(x > -1.0e9 && x < 1.0e9)) return (xf - xf) + 1.0f; /* NaN for NaN/Inf, 1 for huge */
    n = om_rem_pio2(x, &r);
    switch (n & 3) {
    case 0: v = om_cos_k(r); break;
    case 1: v = -om_sin_k(r); break;
    case 2: v = -om_cos_k(r); break;
    default: v = om_sin_k(r); break;
    }
    return (float)v;
}

/* ort_sinf(x) and ort_cosf(x) of the same argument in one go: the reduction and each of the two kernels are
   evaluated once and the quadrant only selects and negates (exact), so both results have the bits of the two
   separate functions -- without their eight divergent branch bodies on a GPU wave. */
ORT_HD inline void ort_sincosf(float xf, float *sn, float *cs)
{
    double x = (double)xf, r;
    if (!(x > -1.0e9 && x < 1.0e9)) { *sn = xf - xf; *cs = (xf - xf) + 1.0f; return; }
    const int n = om_rem_pio2(x, &r);
    const double sk = om_sin_k(r), ck = om_cos_k(r);
    double sv = (n & 1) ? ck : sk; /* n & 3: 0 sin_k, 1 cos_k, 2 -sin_k, 3 -cos_k */
    double cv = (n & 1) ? sk : ck; /*        0 cos_k, 1 -sin_k, 2 -cos_k, 3 sin_k */
    if (n & 2) sv = -sv;
    if ((n + 1) & 2) cv = -cv;
    *sn = (float)sv;
    *cs = (float)cv;
}

/* atan(t) for |t| <= 0.4143: alternating Taylor series to t^35. */
ORT_HD inline double om_atan_series(double t)
{
    double z = t * t;
    double p = 1.0 / 35.0;
    p = 1.0 / 33.0 - z * p;
    p = 1.0 / 31.0 - z * p;
    p = 1.0 / 29.0 - z * p;
    p = 1.0 / 27.0 - z * p;
    p = 1.0 / 25.0 - z * p;
    p = 1.0 / 23.0 - z * p;
    p = 1.0 / 21.0 - z * p;
    p = 1.0 / 19.0 - z * p;
    p = 1.0 / 17.0 - z * p;
    p = 1.0 / 15.0 - z * p;
    p = 1.0 / 13.0 - z * p;
    p = 1.0 / 11.0 - z * p;
    p = 1.0 / 9.0 - z * p;
    p = 1.0 / 7.0 - z * p;
    p = 1.0 / 5.0 - z * p;
    p = 1.0 / 3.0 - z * p;
    return t - t * (z * p);
}

/* atan(z) for z >= 0 (z may be +inf). */
ORT_HD inline double om_atan_pos(double z)
{
    if (z <= 0.41421356237309503) return om_atan_series(z);
    if (z < 2.4142135623730951) return OM_PI_4 + om_atan_series((z - 1.0) / (z + 1.0));
    return OM_PI_2 + om_atan_series(-1.0 / z);
}

ORT_HD inline float ort_atan2f(float yf, float xf)
{
    uint32_t yb = om_f32_bits(yf), xb = om_f32_bits(xf);
    int yneg = (int)(yb >> 31), xneg = (int)(xb >> 31);
    double ay, ax, a;
    if (om_isnan_f(yf) || om_isnan_f(xf)) return om_nan_f();
    ay = (double)om_bits_f32(yb & 0x7fffffffu);
    ax = (double)om_bits_f32(xb & 0x7fffffffu);
    if (ay == 0.0) {
        a = xneg ? OM_PI : 0.0;
    } else if (ax == 0.0) {
        a = OM_PI_2;
    } else if (om_isinf_f(yf)) {
        a = om_isinf_f(xf) ? (xneg ? 3.0 * OM_PI_4 : OM_PI_4) : OM_PI_2;
    } else if (om_isinf_f(xf)) {
        a = xneg ? OM_PI : 0.0;
    } else {
        a = om_atan_pos(ay / ax);
        if (xneg) a = OM_PI - a;
    }
    return (float)(yneg ? -a : a);
}

/* log(x) for finite x > 0 given as a double that is a NORMAL binary64. */
ORT_HD inline double om_log_pos(double x)
{
    uint64_t b = om_f64_bits(x);
    int e = (int)((b >> 52) & 0x7ff) - 1023;
    double m = om_bits_f64((b & 0x000fffffffffffffull) | 0x3ff0000000000000ull); /* [1,2) */
    double s, z, p;
    if (m > 1.4142135623730951) { m = m * 0.5; e = e + 1; }
    s = (m - 1.0) / (m + 1.0); /* |s| <= 0.1716 */
    z = s * s;
    p = 1.0 / 21.0;
    p = 1.0 / 19.0 + z * p;
    p = 1.0 / 17.0 + z * p;
    p = 1.0 / 15.0 + z * p;
    p = 1.0 / 13.0 + z * p;
    p = 1.0 / 11.0 + z * p;
    p = 1.0 / 9.0 + z * p;
    p = 1.0 / 7.0 + z * p;
    p = 1.0 / 5.0 + z * p;
    p = 1.0 / 3.0 + z * p;
    p = 2.0 * s + 2.0 * s * (z * p);
    return ((double)e * OM_LN2_HI + p) + (double)e * OM_LN2_LO;
}

ORT_HD inline float ort_logf(float xf)
{
    if (om_isnan_f(xf)) return om_nan_f();
    if (xf == 0.0f) return -om_inf_f();
    if (xf < 0.0f) return om_nan_f();
    if (om_isinf_f(xf)) return xf;
    return (float)om_log_pos((double)xf);
}

/* exp(v), |v| <= 120, as a double. */
ORT_HD inline double om_exp(double v)
{
    double fk = v * OM_INV_LN2;
    int k = (int)(fk + (fk < 0.0 ? -0.5 : 0.5));
    double dk = (double)k;
    double r = (v - dk * OM_LN2_HI) - dk * OM_LN2_LO; /* |r| <= 0.3466 */
    double p = 1.0 / 6227020800.0; /* 1/13! */
    p = 1.0 / 479001600.0 + r * p;
    p = 1.0 / 39916800.0 + r * p;
    p = 1.0 / 3628800.0 + r * p;
    p = 1.0 / 362880.0 + r * p;
    p = 1.0 / 40320.0 + r * p;
    p = 1.0 / 5040.0 + r * p;
    p = 1.0 / 720.0 + r * p;
    p = 1.0 / 120.0 + r * p;
    p = 1.0 / 24.0 + r * p;
    p = 1.0 / 6.0 + r * p;
    p = 0.5 + r * p;
    p = 1.0 + r * p;
    p = 1.0 + r * p;
    return p * om_bits_f64((uint64_t)(k + 1023) << 52);
}

/* C99 F.9.4.4 special cases, then exp(y*log|x|) in binary64. */
ORT_HD inline float ort_powf(float xf, float yf)
{
    uint32_t xb = om_f32_bits(xf), yb = om_f32_bits(yf);
    uint32_t ax = xb & 0x7fffffffu, ay = yb & 0x7fffffffu;
    int y_is_int = 0, y_is_odd = 0;
    double v, r;
    if (ay == 0u) return 1.0f;
    if (xb == 0x3f800000u) return 1.0f;
    if (ax > 0x7f800000u || ay > 0x7f800000u) return om_nan_f();
    if (ay >= 0x4b800000u) { /* |y| >= 2^24 (or inf): an even integer */
        y_is_int = (ay != 0x7f800000u);
    } else if (ay >= 0x3f800000u) {
        int yi = (int)yf;
        if ((float)yi == yf) { y_is_int = 1; y_is_odd = yi & 1; }
    }
    if (ay == 0x7f800000u) { /* y = +-inf */
        if (ax == 0x3f800000u) return 1.0f;
        if ((ax > 0x3f800000u) == ((yb >> 31) == 0u)) return om_inf_f();
        return 0.0f;
    }
    if (ax == 0u || ax == 0x7f800000u) { /* x = +-0 or +-inf */
        int big = (ax != 0u) == ((yb >> 31) == 0u); /* result magnitude is inf */
        float mag = big ? om_inf_f() : 0.0f;
        return ((xb >> 31) && y_is_odd) ? -mag : mag;
    }
    if (xb >> 31) {
        if (!y_is_int) return om_nan_f();
    }
    v = (double)yf * om_log_pos((double)om_bits_f32(ax));
    if (v > 100.0) r = (double)om_inf_f();
    else if (v < -120.0) r = 0.0;
    else r = om_exp(v);
    if ((xb >> 31) && y_is_odd) r = -r;
    return (float)r;
}


#endif /* ORT_DETMATH_H */
