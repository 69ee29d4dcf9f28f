/*
 * oracle/det_math.h -- TEST INFRASTRUCTURE (oracle side). Not part of the product.
 *
 * Deterministic replacements for the libm calls on the reference's hot path
 * (sinf cosf atan2f powf logf: /root/reference/code/ray.cpp:829,857,964-966,1072,
 * 1128,1138,1233-1234; random.h:107-110; math.h:730-733,754-757 for the mesh
 * placement; parser.cpp:247 for exponents).
 *
 * Why: the reference's results depend on whichever libm it is linked against
 * (Apple libm originally, glibc 2.35 in this container), and no reference test
 * pins those bits.  A GPU cannot reproduce glibc's bits, so the parity anchor is
 * "reference sources + this libm": every function below uses only IEEE-754
 * binary64 + - * / and integer ops (no fma, no sqrt, no tables), so it yields the
 * same bits on x86-64 and on gfx950 when compiled with -ffp-contract=off.
 * Results are within ~1e-13 relative of the true value before the final
 * round-to-float, i.e. correctly rounded f32 except in ~1e-6 of cases.
 *
 * The same algorithm is restated for the device in
 * offline_raytracer_amd/csrc/ort_detmath.h; tests/test_oracle_golden.py::test_libm_is_close_to_glibc and the unit tables (op 9) check the two
 * bit for bit.
 */
#ifndef ORT_ORACLE_DET_MATH_H
#define ORT_ORACLE_DET_MATH_H

#include <stdint.h>
#include <string.h>

static inline uint64_t dm_f64_bits(double d) { uint64_t u; memcpy(&u, &d, 8); return u; }
static inline double dm_bits_f64(uint64_t u) { double d; memcpy(&d, &u, 8); return d; }
static inline uint32_t dm_f32_bits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float dm_bits_f32(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

static inline int dm_isnan_f(float x) { return (dm_f32_bits(x) & 0x7fffffffu) > 0x7f800000u; }
static inline int dm_isinf_f(float x) { return (dm_f32_bits(x) & 0x7fffffffu) == 0x7f800000u; }
static inline float dm_nan_f(void) { return dm_bits_f32(0x7fc00000u); }
static inline float dm_inf_f(void) { return dm_bits_f32(0x7f800000u); }

#define DM_PIO2_HI 1.57079632673412561417e+00 /* first 33 bits of pi/2 */
#define DM_PIO2_LO 6.07710050650619224932e-11 /* pi/2 - DM_PIO2_HI */
#define DM_INV_PIO2 6.36619772367581382433e-01
#define DM_PI 3.14159265358979311600e+00
#define DM_PI_2 1.57079632679489655800e+00
#define DM_PI_4 7.85398163397448278999e-01
#define DM_LN2_HI 6.93147180369123816490e-01
#define DM_LN2_LO 1.90821492927058770002e-10
#define DM_INV_LN2 1.44269504088896338700e+00

/* sin(r), cos(r) for |r| <= pi/4 (+ a little); minimax coefficients as published
 * in FreeBSD msun k_sin.c / k_cos.c. */
static inline double dm_sin_k(double r)
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

static inline double dm_cos_k(double r)
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
static inline int dm_rem_pio2(double x, double *r)
{
    double fn = x * DM_INV_PIO2;
    int n = (int)(fn + (fn < 0.0 ? -0.5 : 0.5));
    double dn = (double)n;
    *r = (x - dn * DM_PIO2_HI) - dn * DM_PIO2_LO;
    return n;
}

static inline float det_sinf(float xf)
{
    double x = (double)xf, r, v;
    int n;
    if (!(x > -1.0e9 && x < 1.0e9)) return xf - xf; /* NaN for NaN/Inf, 0 for huge */
    n = dm_rem_pio2(x, &r);
    switch (n & 3) {
    case 0: v = dm_sin_k(r); break;
    case 1: v = dm_cos_k(r); break;
    case 2: v = -dm_sin_k(r); break;
    default: v = -dm_cos_k(r); break;
    }
    return (float)v;
}

static inline float det_cosf(float xf)
{
    double x = (double)xf, r, v;
    int n;
    if (!(x > -1.0e9 && x < 1.0e9)) return (xf - xf) + 1.0f; /* NaN for NaN/Inf, 1 for huge */
    n = dm_rem_pio2(x, &r);
    switch (n & 3) {
    case 0: v = dm_cos_k(r); break;
    case 1: v = -dm_sin_k(r); break;
    case 2: v = -dm_cos_k(r); break;
    default: v = dm_sin_k(r); break;
    }
    return (float)v;
}

/* atan(t) for |t| <= 0.4143: alternating Taylor series to t^35. */
static inline double dm_atan_series(double t)
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
static inline double dm_atan_pos(double z)
{
    if (z <= 0.41421356237309503) return dm_atan_series(z);
    if (z < 2.4142135623730951) return DM_PI_4 + dm_atan_series((z - 1.0) / (z + 1.0));
    return DM_PI_2 + dm_atan_series(-1.0 / z);
}

static inline float det_atan2f(float yf, float xf)
{
    uint32_t yb = dm_f32_bits(yf), xb = dm_f32_bits(xf);
    int yneg = (int)(yb >> 31), xneg = (int)(xb >> 31);
    double ay, ax, a;
    if (dm_isnan_f(yf) || dm_isnan_f(xf)) return dm_nan_f();
    ay = (double)dm_bits_f32(yb & 0x7fffffffu);
    ax = (double)dm_bits_f32(xb & 0x7fffffffu);
    if (ay == 0.0) {
        a = xneg ? DM_PI : 0.0;
    } else if (ax == 0.0) {
        a = DM_PI_2;
    } else if (dm_isinf_f(yf)) {
        a = dm_isinf_f(xf) ? (xneg ? 3.0 * DM_PI_4 : DM_PI_4) : DM_PI_2;
    } else if (dm_isinf_f(xf)) {
        a = xneg ? DM_PI : 0.0;
    } else {
        a = dm_atan_pos(ay / ax);
        if (xneg) a = DM_PI - a;
    }
    return (float)(yneg ? -a : a);
}

/* log(x) for finite x > 0 given as a double that is a NORMAL binary64. */
static inline double dm_log_pos(double x)
{
    uint64_t b = dm_f64_bits(x);
    int e = (int)((b >> 52) & 0x7ff) - 1023;
    double m = dm_bits_f64((b & 0x000fffffffffffffull) | 0x3ff0000000000000ull); /* [1,2) */
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
    return ((double)e * DM_LN2_HI + p) + (double)e * DM_LN2_LO;
}

static inline float det_logf(float xf)
{
    if (dm_isnan_f(xf)) return dm_nan_f();
    if (xf == 0.0f) return -dm_inf_f();
    if (xf < 0.0f) return dm_nan_f();
    if (dm_isinf_f(xf)) return xf;
    return (float)dm_log_pos((double)xf);
}

/* exp(v), |v| <= 120, as a double. */
static inline double dm_exp(double v)
{
    double fk = v * DM_INV_LN2;
    int k = (int)(fk + (fk < 0.0 ? -0.5 : 0.5));
    double dk = (double)k;
    double r = (v - dk * DM_LN2_HI) - dk * DM_LN2_LO; /* |r| <= 0.3466 */
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
    return p * dm_bits_f64((uint64_t)(k + 1023) << 52);
}

/* C99 F.9.4.4 special cases, then exp(y*log|x|) in binary64. */
static inline float det_powf(float xf, float yf)
{
    uint32_t xb = dm_f32_bits(xf), yb = dm_f32_bits(yf);
    uint32_t ax = xb & 0x7fffffffu, ay = yb & 0x7fffffffu;
    int y_is_int = 0, y_is_odd = 0;
    double v, r;
    if (ay == 0u) return 1.0f;
    if (xb == 0x3f800000u) return 1.0f;
    if (ax > 0x7f800000u || ay > 0x7f800000u) return dm_nan_f();
    if (ay >= 0x4b800000u) { /* |y| >= 2^24 (or inf): an even integer */
        y_is_int = (ay != 0x7f800000u);
    } else if (ay >= 0x3f800000u) {
        int yi = (int)yf;
        if ((float)yi == yf) { y_is_int = 1; y_is_odd = yi & 1; }
    }
    if (ay == 0x7f800000u) { /* y = +-inf */
        if (ax == 0x3f800000u) return 1.0f;
        if ((ax > 0x3f800000u) == ((yb >> 31) == 0u)) return dm_inf_f();
        return 0.0f;
    }
    if (ax == 0u || ax == 0x7f800000u) { /* x = +-0 or +-inf */
        int big = (ax != 0u) == ((yb >> 31) == 0u); /* result magnitude is inf */
        float mag = big ? dm_inf_f() : 0.0f;
        return ((xb >> 31) && y_is_odd) ? -mag : mag;
    }
    if (xb >> 31) {
        if (!y_is_int) return dm_nan_f();
    }
    v = (double)yf * dm_log_pos((double)dm_bits_f32(ax));
    if (v > 100.0) r = (double)dm_inf_f();
    else if (v < -120.0) r = 0.0;
    else r = dm_exp(v);
    if ((xb >> 31) && y_is_odd) r = -r;
    return (float)r;
}

#endif /* ORT_ORACLE_DET_MATH_H */
