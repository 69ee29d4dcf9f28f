/*
 * oracle/ort_oracle.c -- TEST INFRASTRUCTURE.  CPU restatement (plain C99) of the
 * reference's per-pixel path-trace hot path.  Never linked into, imported by or called
 * from the product; see ort_oracle.h.
 *
 * Every function names the reference lines it follows (paths relative to
 * /root/reference/code).  Build with -ffp-contract=off: the reference's own output
 * changes under FMA contraction.  libm -> det_math.h.
 *
 * PINNED: bit-identical to the reference compiled from its own sources (oracle/_ref/ref_det, recipe in
 * oracle/Makefile) -- images, the reference's shapes_tested counter and final RNG state in all seeding
 * policies, closest-hit tables, per-function tables: tests/golden/ (tests/test_oracle_golden.py), and
 * 1 343 further renders on fresh seeds (tools/oracle_vs_ref.py, profiles/r01_stress_parity.md).
 */
#include "ort_oracle.h"
#include "det_math.h"

#include <math.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <float.h>

/* ---- scalar helpers with the reference's exact macro semantics ------------------- */
#define O_MIN(a, b) (((a) < (b)) ? (a) : (b))             /* types.h:51 */
#define O_MAX(a, b) (((a) > (b)) ? (a) : (b))             /* types.h:50 */
#define O_SIGN(a) (((a) >= 0.0f) ? 1.0f : -1.0f)          /* types.h:52 */
#define O_PI 3.14159265358979323846264338327950288419716939937510582097494459230f /* platform.h:44 */
#define O_EULER 2.71828182845904523536028747135266249f    /* ray.cpp:4 */
#define O_HIT_T_MIN 0.000001f                             /* ray.cpp:5 */

static float o_abs(float v) { if (v <= 0.0f) v *= -1.0f; return v; }      /* intrinsic.h:132-143 */
static float o_sq(float v) { return v * v; }                               /* intrinsic.h:145-151 */
static int o_ceq(float a, float b)                                         /* math.h:9-22 */
{
    float tol = 0.000001f, diff = a - b;
    return (diff >= -tol && diff < tol);
}

static o_v3 v3(float x, float y, float z) { o_v3 r; r.x = x; r.y = y; r.z = z; return r; }
static o_v3 v_add(o_v3 a, o_v3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }           /* math.h:210-221 */
static o_v3 v_sub(o_v3 a, o_v3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }           /* math.h:223-233 */
static o_v3 v_neg(o_v3 a) { return v3(-a.x, -a.y, -a.z); }                                  /* math.h:197-208 */
static o_v3 v_scale(float s, o_v3 a) { return v3(s * a.x, s * a.y, s * a.z); }              /* math.h:266-276 */
static o_v3 v_div(o_v3 a, float s) { return v3(a.x / s, a.y / s, a.z / s); }                /* math.h:234-244 */
static o_v3 v_had(o_v3 a, o_v3 b) { return v3(a.x * b.x, a.y * b.y, a.z * b.z); }           /* math.h:325-329 */
static float v_dot(o_v3 a, o_v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }            /* math.h:319-323 */
static float v_len2(o_v3 a) { return a.x * a.x + a.y * a.y + a.z * a.z; }                   /* math.h:292-296 */
static float v_len(o_v3 a) { return sqrtf(a.x * a.x + a.y * a.y + a.z * a.z); }             /* math.h:173-177 */
static o_v3 v_cross(o_v3 a, o_v3 b)                                                         /* math.h:280-290 */
{
    return v3(a.y * b.z - b.y * a.z, b.x * a.z - a.x * b.z, a.x * b.y - b.x * a.y);
}
static o_v3 v_normalize(o_v3 a)                                                             /* math.h:298-310 */
{
    float l = v_len(a);
    if (!o_ceq(l, 0.0f)) return v_div(a, l);
    return v3(0, 0, 0);
}
static int v_is0(o_v3 v)                                                                    /* math.h:331-345 */
{
    float tol = 0.000001f;
    return (v.x >= -tol && v.x < tol && v.y >= -tol && v.y < tol && v.z >= -tol && v.z < tol);
}
static int v_isnan(o_v3 v) { return isnan(v.x) || isnan(v.y) || isnan(v.z); }               /* math.h:361-371 */
static int v_isinf(o_v3 v) { return isinf(v.x) || isinf(v.y) || isinf(v.z); }               /* math.h:373-383 */
static o_v3 v_min(o_v3 a, o_v3 b) { return v3(O_MIN(a.x, b.x), O_MIN(a.y, b.y), O_MIN(a.z, b.z)); } /* math.h:1084-1094 */
static o_v3 v_max(o_v3 a, o_v3 b) { return v3(O_MAX(a.x, b.x), O_MAX(a.y, b.y), O_MAX(a.z, b.z)); } /* math.h:1096-1106 */
static int v_in_rect(o_v3 p, o_v3 lo, o_v3 hi)                                              /* math.h:1156-1169 */
{
    return (p.x >= lo.x && p.x < hi.x) && (p.y >= lo.y && p.y < hi.y) && (p.z >= lo.z && p.z < hi.z);
}

typedef struct { o_v3 r0, r1, r2; } o_m3; /* row major, types.h:138-156 */
static o_v3 m3_mul(o_m3 m, o_v3 v) { return v3(v_dot(m.r0, v), v_dot(m.r1, v), v_dot(m.r2, v)); } /* math.h:958-968 */
static o_m3 m3_transpose(o_m3 m)                                                                /* math.h:983-997 */
{
    o_m3 r;
    r.r0 = v3(m.r0.x, m.r1.x, m.r2.x);
    r.r1 = v3(m.r0.y, m.r1.y, m.r2.y);
    r.r2 = v3(m.r0.z, m.r1.z, m.r2.z);
    return r;
}

/* ---- RNG (random.h) ---------------------------------------------------------------- */
static void rng_step(uint32_t *s)                     /* random.h:5-15; third shift is RIGHT */
{
    uint32_t x = *s;
    x ^= x << 13;
    x ^= x >> 17;
    x ^= x >> 5;
    *s = x;
}
static float rng_01(uint32_t *s)                      /* random.h:31-38 */
{
    rng_step(s);
    return (float)*s / (float)UINT32_MAX;
}
static float rng_between(uint32_t *s, float lo, float hi) /* random.h:47-53: advances twice */
{
    rng_step(s);
    return lo + (hi - lo) * rng_01(s);
}
static uint32_t rng_between_u32(uint32_t *s, uint32_t lo, uint32_t one_past) /* random.h:75-81 */
{
    rng_step(s);
    return (uint32_t)(*s % (one_past - lo) + lo);
}
static uint32_t rng_u32(uint32_t *s) { rng_step(s); return *s; } /* random.h:83-89 */
static o_v3 rng_spherical(uint32_t *s, float phi_min, float phi_max, float th_min, float th_max) /* random.h:100-117 */
{
    float phi = rng_between(s, phi_min, phi_max);
    float theta = rng_between(s, th_min, th_max);
    float sp = det_sinf(phi), cp = det_cosf(phi), st = det_sinf(theta), ct = det_cosf(theta);
    return v3(cp * ct, cp * st, sp);
}

static uint32_t fmix32(uint32_t h)
{
    h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16;
    return h;
}
uint32_t oracle_job_seed(uint32_t master, uint32_t job)
{
    uint32_t h = fmix32(master ^ (job * 2654435761u));
    return h ? h : 1u;
}

/* ---- intersectors -------------------------------------------------------------------- */
typedef struct { float t; o_v3 n; int32_t inner; } o_hit;

static o_hit hit_triangle(o_v3 v0, o_v3 v1, o_v3 v2, o_v3 o, o_v3 d)      /* ray.cpp:63-115 */
{
    o_hit r; r.t = -1.0f; r.n = v3(0, 0, 0); r.inner = 0;
    o_v3 e1 = v_sub(v1, v0), e2 = v_sub(v2, v0);
    o_v3 pv = v_cross(d, e2);
    float det = v_dot(pv, e1);
    o_v3 T = v_sub(o, v0);
    float tol = 0.000001f;
    if (det <= -tol || det >= tol) {
        o_v3 a = v_cross(T, e1);
        float t = v_dot(a, e2) / det;
        float u = v_dot(pv, T) / det;
        float v = v_dot(a, d) / det;
        if (t >= O_HIT_T_MIN && u >= 0.0f && v >= 0.0f && u + v <= 1.0f) {
            r.t = t;
            r.n = v_cross(e1, e2);
        }
    }
    return r;
}

static o_hit hit_sphere(o_v3 c, float rad, o_v3 o, o_v3 d)                  /* ray.cpp:132-190 */
{
    o_hit r; r.t = -1.0f; r.n = v3(0, 0, 0); r.inner = 0;
    o_v3 rel = v_sub(o, c);
    float a = v_dot(d, d), b = v_dot(d, rel), cc = v_dot(rel, rel) - rad * rad;
    float root = b * b - a * cc;
    float tol = 0.00001f;
    if (root >= tol) {
        float sq = sqrtf(root);
        float tn = (-b - sq) / a, tp = (-b + sq) / a;
        float hit_normal_c = 1.0f, t;
        if (tn < 0.0f) { t = tp; r.inner = 1; } else { t = tn; }
        if (t > O_HIT_T_MIN) {
            r.t = t;
            r.n = v_scale(hit_normal_c, v_sub(v_add(o, v_scale(r.t, d)), c));
        }
    } else if (root < tol && root > -tol) {
        float t = (-b) / (2 * a);
        if (t > O_HIT_T_MIN) {
            r.t = t;
            r.n = v_sub(v_add(o, v_scale(r.t, d)), c);
        }
    }
    return r;
}

static o_hit hit_aab(o_v3 lo, o_v3 hi, o_v3 o, o_v3 d)                       /* ray.cpp:206-283 */
{
    o_hit r; r.t = -1.0f; r.n = v3(0, 0, 0); r.inner = 0;
    o_v3 inv = v3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
    o_v3 t0 = v_had(v_sub(lo, o), inv), t1 = v_had(v_sub(hi, o), inv);
    o_v3 tmin = v_min(t0, t1), tmax = v_max(t0, t1);
    float max_of_min = O_MAX(O_MAX(tmin.x, tmin.y), tmin.z); /* math.h:1068-1074 */
    float min_of_max = O_MIN(O_MIN(tmax.x, tmax.y), tmax.z); /* math.h:1076-1082 */
    if (min_of_max >= max_of_min) {
        float tx = t0.x, ty = t0.y, tz = t0.z, best;
        o_v3 nx = v3(-1, 0, 0), ny = v3(0, -1, 0), nz = v3(0, 0, -1), bn;
        if (tx > t1.x) { tx = t1.x; nx = v3(1, 0, 0); }
        if (ty > t1.y) { ty = t1.y; ny = v3(0, 1, 0); }
        if (tz > t1.z) { tz = t1.z; nz = v3(0, 0, 1); }
        best = tx; bn = nx;
        if (best < ty) { best = ty; bn = ny; }
        if (best < tz) { best = tz; bn = nz; }
        r.t = max_of_min;
        r.n = bn;
    }
    return r;
}

static o_m3 rot_along_z(o_v3 src)                                           /* ray.cpp:8-33 */
{
    o_m3 m;
    m.r0 = v3(1, 0, 0); m.r1 = v3(0, 1, 0); m.r2 = v3(0, 0, 1);
    if (!v_is0(v_cross(src, v3(0, 0, 1)))) {
        o_v3 a = v_normalize(src);
        o_v3 b = v_normalize(v_cross(v3(0, 0, 1), a));
        o_v3 c;
        if (v_is0(b)) b = v_normalize(v_cross(v3(1, 0, 0), a));
        c = v_cross(a, b);
        m.r0 = b; m.r1 = c; m.r2 = a;
    }
    return m;
}

static o_hit hit_cylinder(o_v3 base, o_v3 axis, float radius, o_v3 o, o_v3 d) /* ray.cpp:286-352 */
{
    o_hit r; r.t = -1.0f; r.n = v3(0, 0, 0); r.inner = 0;
    o_m3 rot = rot_along_z(axis);
    float t_bot, t_top, smin, smax, a, b, c, det;
    o = m3_mul(rot, v_sub(o, base));
    d = m3_mul(rot, d);
    t_bot = (-o.z) / d.z;
    t_top = (v_len(axis) - o.z) / d.z;
    smin = O_MIN(t_bot, t_top);
    smax = O_MAX(t_bot, t_top);
    a = d.x * d.x + d.y * d.y;            /* math.h:dot(v2) */
    b = d.x * o.x + d.y * o.y;
    c = (o.x * o.x + o.y * o.y) - radius * radius;
    det = b * b - a * c;
    if (det >= 0.0f) {
        float sq = sqrtf(det);
        float cmin = (-b - sq) / a, cmax = (-b + sq) / a;
        float tin = O_MAX(smin, cmin), tout = O_MIN(smax, cmax);
        if (tin <= tout) {
            r.t = tin;
            r.n = v3(0, 1, 0);
            if (smin < cmin) {
                o_v3 p = v_add(o, v_scale(r.t, d));
                r.n = v3(p.x, p.y, 0);
            }
            r.n = m3_mul(m3_transpose(rot), r.n);
        }
    }
    return r;
}

/* ---- BSDF (ray.cpp:825-1161) ------------------------------------------------------- */
static o_v3 bsdf_fresnel(o_v3 Ks, float l_dot_h)                              /* ray.cpp:825-831 */
{
    float k = 1 - det_powf(1.0f - o_abs(l_dot_h), 5.0f);
    return v_add(Ks, v_scale(k, v_sub(v3(1, 1, 1), Ks)));
}

static float bsdf_ggx(o_v3 N, o_v3 H, float rough)                            /* ray.cpp:834-865 */
{
    float result = 0.0f, ndh = v_dot(N, H);
    if (ndh > 0.0f) {
        float r2 = o_sq(rough);
        float tan_t = sqrtf(1.0f - o_sq(ndh)) / ndh;
        float denom = O_PI * det_powf(ndh, 4.0f) * o_sq(r2 + o_sq(tan_t));
        if (!o_ceq(denom, 0.0f)) result = r2 / denom;
    }
    return result;
}

static float bsdf_geometry(o_v3 w, o_v3 N, o_v3 m, float rough)               /* ray.cpp:868-897 */
{
    float result = 0.0f, wdn = v_dot(w, N), wdm = v_dot(w, m);
    if (!o_ceq(wdm, 0.0f) && (wdn / wdm) > 0) {
        if (wdm > 1.0f) {
            result = 1.0f;
        } else {
            float tan_t = sqrtf(1.0f - o_sq(wdn)) / wdn;
            if (!o_ceq(tan_t, 0.0f)) {
                float r2 = o_sq(rough);
                result = 2.0f / (1.0f + sqrtf(1 + r2 * o_sq(tan_t)));
            }
        }
    }
    return result;
}

static float bsdf_radicand(o_v3 m, o_v3 wo, float n)                          /* ray.cpp:899-904 */
{
    return 1 - o_sq(n) * (1 - o_sq(v_dot(wo, m)));
}

typedef struct { float ni, no, n; } o_beer;
static o_beer bsdf_beer(o_v3 N, o_v3 wo, float ior)                           /* ray.cpp:914-933 */
{
    o_beer r;
    if (v_dot(N, wo) >= 0.0f) { r.ni = 1.0f; r.no = ior; } else { r.ni = ior; r.no = 1.0f; }
    r.n = r.ni / r.no;
    return r;
}

static o_v3 bsdf_eval(o_v3 N, o_v3 wi, o_v3 wo, o_v3 Kd, o_v3 Ks, o_v3 Kt, float ior, float rough, float dist) /* ray.cpp:936-1005 */
{
    o_v3 Ed = v_div(Kd, O_PI);
    o_v3 H = v_scale(O_SIGN(v_dot(wi, N)), v_normalize(v_add(wo, wi)));
    float wi_h = v_dot(wi, H);
    o_v3 Es = v3(0, 0, 0), Et = v3(0, 0, 0);
    float wi_n = v_dot(wi, N), wo_n = v_dot(wo, N);
    if (wi_h > 0.0f && v_len2(Ks) > 0.0f) {
        o_v3 F = bsdf_fresnel(Ks, wi_h);
        float D = bsdf_ggx(N, H, rough);
        float G = bsdf_geometry(wi, N, H, rough) * bsdf_geometry(wo, N, H, rough);
        Es = v_scale((D * G) / (4.0f * o_abs(wi_n) * o_abs(wo_n)), F);
    }
    if (v_len2(Kt) > 0.0f) {
        o_v3 At = v3(1, 1, 1), m;
        o_beer bn;
        float r;
        if (wo_n < 0) {
            At.x = det_powf(O_EULER, dist * det_logf(Kt.x));
            At.y = det_powf(O_EULER, dist * det_logf(Kt.y));
            At.z = det_powf(O_EULER, dist * det_logf(Kt.z));
        }
        bn = bsdf_beer(N, wo, ior);
        m = v_normalize(v_neg(v_add(v_scale(bn.ni, wi), v_scale(bn.no, wo))));
        r = bsdf_radicand(m, wo, bn.n);
        if (r < 0.0f) {
            if (v_len2(Ks) > 0.0f) Et = v_had(At, Es);
        } else {
            float wi_m = v_dot(wi, m), wo_m = v_dot(wo, m);
            o_v3 F = v_sub(v3(1, 1, 1), bsdf_fresnel(Ks, wi_m));
            float D = bsdf_ggx(N, m, rough);
            float G = bsdf_geometry(wi, N, m, rough) * bsdf_geometry(wo, N, m, rough);
            float denom = (o_abs(wi_n) * o_abs(wo_n) * o_sq(bn.ni * wi_m + bn.no * wo_m));
            if (!o_ceq(denom, 0.0f)) {
                o_v3 nom = v_scale(D * G * o_abs(wi_m) * o_abs(wo_m) * o_sq(bn.no), F);
                Et = v_had(At, v_div(nom, denom));
            }
        }
    }
    return v_scale(o_abs(wi_n), v_add(v_add(Ed, Es), Et));
}

static float bsdf_pdf(o_v3 N, o_v3 wi, o_v3 wo, float rough, o_v3 Kd, o_v3 Ks, o_v3 Kt, float ior) /* ray.cpp:1007-1063 */
{
    float kd = v_len(Kd), ks = v_len(Ks), kt = v_len(Kt);
    float s = kd + ks + kt;
    float pd_c = kd / s, ps_c = ks / s, pt_c = kt / s;
    float pd = o_abs(v_dot(wi, N)) / O_PI;
    o_v3 H = v_scale(O_SIGN(v_dot(N, wi)), v_normalize(v_add(wo, wi)));
    float n_h = v_dot(N, H), wi_h = v_dot(wi, H);
    float ps = 0.0f, pt, r;
    o_beer bn;
    o_v3 m;
    if (ps_c > 0.0f) {
        float denom = (4.0f * o_abs(wi_h));
        if (!o_ceq(denom, 0.0f)) {
            float D = bsdf_ggx(N, H, rough);
            ps = D * o_abs(n_h) / denom;
        }
    }
    bn = bsdf_beer(N, wo, ior);
    m = v_normalize(v_neg(v_add(v_scale(bn.ni, wi), v_scale(bn.no, wo))));
    r = bsdf_radicand(m, wo, bn.n);
    pt = ps;
    if (pt_c > 0.0f && r >= 0.0f) {
        float n_m = v_dot(N, m), wi_m = v_dot(wi, m), wo_m = v_dot(wo, m);
        float denom = o_sq(bn.no * wo_m + bn.no * wo_m); /* sic, ray.cpp:1054 */
        if (!o_ceq(denom, 0.0f)) {
            float D = bsdf_ggx(N, m, rough);
            pt = D * o_abs(n_m) * o_sq(bn.no) * o_abs(wi_m) / denom;
        }
    }
    return pd_c * pd + ps_c * ps + pt_c * pt;
}

static o_v3 bsdf_sample_lobe(o_v3 N, float c, float phi)                      /* ray.cpp:1065-1091 */
{
    float s;
    o_v3 K;
    N = v_normalize(N);
    s = sqrtf(1.0f - c * c);
    K = v3(s * det_cosf(phi), s * det_sinf(phi), c);
    if (o_abs(N.z - 1.0f) < 0.0001f) return K;
    if (o_abs(N.z + 1.0f) < 0.0001f) return v3(K.x, -K.y, -K.z);
    {
        o_v3 B = v_normalize(v3(-N.y, N.x, 0));
        o_v3 C = v_cross(N, B);
        return v_add(v_add(v_scale(K.x, B), v_scale(K.y, C)), v_scale(K.z, N));
    }
}

static o_v3 bsdf_sample(uint32_t *rng, o_v3 N, o_v3 wo, float rough, o_v3 Kd, o_v3 Ks, o_v3 Kt, float ior, int *is_trans) /* ray.cpp:1100-1161 */
{
    float kd = v_len(Kd), ks = v_len(Ks), kt = v_len(Kt);
    float s = kd + ks + kt;
    float pd_c = kd / s, ps_c = ks / s;
    float e0 = rng_01(rng), e1 = rng_01(rng), choice = rng_01(rng);
    o_v3 wi;
    *is_trans = 0;
    if (choice < pd_c) {
        wi = bsdf_sample_lobe(N, sqrtf(e0), 2.0f * O_PI * e1);
    } else if (choice >= pd_c && choice < pd_c + ps_c) {
        float ct = det_cosf(det_atan2f(rough * sqrtf(e0), sqrtf(1.0f - e0)));
        o_v3 m = bsdf_sample_lobe(N, ct, 2.0f * O_PI * e1);
        wi = v_sub(v_scale(2.0f * o_abs(v_dot(wo, m)), m), wo);
    } else {
        float ct = det_cosf(det_atan2f(rough * sqrtf(e0), sqrtf(1.0f - e0)));
        o_v3 m = bsdf_sample_lobe(N, ct, 2.0f * O_PI * e1);
        o_beer bn = bsdf_beer(N, wo, ior);
        float r = bsdf_radicand(m, wo, bn.n);
        if (r < 0.0f) {
            wi = v_sub(v_scale(2.0f * o_abs(v_dot(wo, m)), m), wo);
        } else {
            wi = v_sub(v_scale(bn.n * v_dot(wo, m) - O_SIGN(v_dot(wo, N)) * sqrtf(r), m), v_scale(bn.n, wo));
            *is_trans = 1;
        }
    }
    (void)kt;
    return v_normalize(wi);
}

/* ---- scene + the reference's loose octree (ray.cpp:1468-2045) ------------------------ */
enum { T_SPHERE = 1, T_CYL = 2, T_AAB = 3, T_MESH = 4, T_TRI = 5, T_CSG = 6 }; /* ray.h:97-106 */
static const uint32_t k_rec_bytes[7] = { 0, 4 + 20, 4 + 32, 4 + 28, 0, 4 + 24, 4 + 76 }; /* PROBE sizes, SURVEY 8a */

typedef struct { uint8_t type; uint32_t index; } o_rec;
typedef struct {
    int32_t first_child; /* -1: none */
    int32_t is_leaf;
    o_rec *recs;
    uint32_t nrecs, cap;
    o_v3 lo, hi;
} o_node;

typedef struct { uint32_t mesh, i0, i1, i2; } o_tri;

struct o_scene {
    o_material *materials; uint32_t material_count;
    o_sphere *spheres; uint32_t sphere_count;
    o_box *boxes; uint32_t box_count;
    o_cylinder *cylinders; uint32_t cylinder_count;
    o_mesh *meshes; uint32_t mesh_count;
    o_light *lights; uint32_t light_count;
    o_tri *tris; uint32_t tri_count;
    /* CSG stand-in: only its AABB matters */
    int has_csg; o_v3 csg_lo, csg_hi;
    o_node *nodes; uint32_t node_count, node_cap;
    uint32_t depth_limit;
};

typedef struct { o_v3 center, half; } o_aabb;

static o_v3 tri_vertex(const o_scene *s, uint32_t mesh, uint32_t i)
{
    const float *p = s->meshes[mesh].vertices + 3 * (size_t)i;
    return v3(p[0], p[1], p[2]);
}

static o_aabb shape_aabb(const o_scene *s, uint8_t type, uint32_t index) /* ray.cpp:1675-1746 */
{
    o_aabb r;
    o_v3 lo, hi;
    switch (type) {
    case T_SPHERE:
        r.center = s->spheres[index].center;
        r.half = v_scale(s->spheres[index].r, v3(1, 1, 1));
        return r;
    case T_CYL: {
        const o_cylinder *c = &s->cylinders[index];
        o_v3 other = v_add(c->base, c->axis);
        o_v3 q = v_div(v_had(c->axis, c->axis), v_dot(c->axis, c->axis));
        o_v3 e = v_scale(c->r, v_sub(v3(1, 1, 1), v3(sqrtf(q.x), sqrtf(q.y), sqrtf(q.z))));
        lo = v_min(v_sub(c->base, e), v_sub(other, e));
        hi = v_max(v_add(c->base, e), v_add(other, e));
    } break;
    case T_AAB: lo = s->boxes[index].min; hi = s->boxes[index].max; break;
    case T_MESH: lo = s->meshes[index].aabb_min; hi = s->meshes[index].aabb_max; break;
    case T_TRI: {
        const o_tri *t = &s->tris[index];
        o_v3 a = tri_vertex(s, t->mesh, t->i0), b = tri_vertex(s, t->mesh, t->i1), c = tri_vertex(s, t->mesh, t->i2);
        lo = v_min(v_min(a, b), c);
        hi = v_max(v_max(a, b), c);
    } break;
    default: lo = s->csg_lo; hi = s->csg_hi; break; /* T_CSG */
    }
    r.center = v_scale(0.5f, v_add(lo, hi));
    r.half = v_sub(hi, r.center);
    return r;
}

static void grow_aabb(const o_scene *s, o_v3 *lo, o_v3 *hi, uint8_t type, uint32_t index) /* ray.cpp:1765-1777 */
{
    o_aabb a = shape_aabb(s, type, index);
    *lo = v_min(*lo, v_sub(a.center, a.half));
    *hi = v_max(*hi, v_add(a.center, a.half));
}

static uint32_t alloc_children(o_scene *s)                                /* ray.cpp:1839-1840,1748-1763 */
{
    uint32_t first = s->node_count, i;
    if (s->node_count + 8 > s->node_cap) {
        s->node_cap = s->node_cap * 2 + 64;
        s->nodes = (o_node *)realloc(s->nodes, sizeof(o_node) * s->node_cap);
    }
    for (i = 0; i < 8; ++i) {
        o_node *n = &s->nodes[first + i];
        memset(n, 0, sizeof(*n));
        n->first_child = -1;
        n->is_leaf = 1;
        n->lo = v3(FLT_MAX, FLT_MAX, FLT_MAX);
        n->hi = v3(FLT_MIN, FLT_MIN, FLT_MIN); /* sic: smallest positive, ray.cpp:1761 */
    }
    s->node_count += 8;
    return first;
}

static void node_push_rec(o_node *n, uint8_t type, uint32_t index)       /* ray.cpp:1524-1629 */
{
    if (n->nrecs == n->cap) {
        n->cap = n->cap ? n->cap * 2 : 4;
        n->recs = (o_rec *)realloc(n->recs, sizeof(o_rec) * n->cap);
    }
    n->recs[n->nrecs].type = type;
    n->recs[n->nrecs].index = index;
    n->nrecs++;
}

/* child slot + child cell for a shape centre: ray.cpp:1476-1522, then the lowest set
   bit of the surviving mask (platform.h:140-162): bit0 = x>=c, bit1 = y>=c, bit2 = z>=c */
static uint32_t child_slot(o_v3 c, o_v3 half, o_v3 p, o_v3 *cc, o_v3 *ch)
{
    uint32_t slot = 0;
    *ch = v_scale(0.5f, half);
    *cc = c;
    if (p.x >= c.x) { slot |= 1; cc->x += ch->x; } else { cc->x -= ch->x; }
    if (p.y >= c.y) { slot |= 2; cc->y += ch->y; } else { cc->y -= ch->y; }
    if (p.z >= c.z) { slot |= 4; cc->z += ch->z; } else { cc->z -= ch->z; }
    return slot;
}

static void push_into_node(o_scene *s, uint32_t ni, o_v3 c, o_v3 half, uint32_t depth, uint8_t type, uint32_t index) /* ray.cpp:1799-1948 */
{
    grow_aabb(s, &s->nodes[ni].lo, &s->nodes[ni].hi, type, index);
    if (depth < s->depth_limit) {
        if (s->nodes[ni].first_child >= 0) {
            o_v3 cc, ch;
            uint32_t slot = child_slot(c, half, shape_aabb(s, type, index).center, &cc, &ch);
            push_into_node(s, (uint32_t)s->nodes[ni].first_child + slot, cc, ch, depth + 1, type, index);
        } else if (s->nodes[ni].nrecs == 0) {
            node_push_rec(&s->nodes[ni], type, index);
        } else {
            uint32_t first = alloc_children(s), k, nold;
            o_rec *old;
            o_v3 cc, ch;
            uint32_t slot;
            s->nodes[ni].first_child = (int32_t)first;
            old = s->nodes[ni].recs;
            nold = s->nodes[ni].nrecs;
            for (k = 0; k < nold; ++k) {
                slot = child_slot(c, half, shape_aabb(s, old[k].type, old[k].index).center, &cc, &ch);
                push_into_node(s, first + slot, cc, ch, depth + 1, old[k].type, old[k].index);
            }
            free(old);
            s->nodes[ni].recs = 0;
            s->nodes[ni].nrecs = 0;
            s->nodes[ni].cap = 0;
            s->nodes[ni].is_leaf = 0;
            slot = child_slot(c, half, shape_aabb(s, type, index).center, &cc, &ch);
            push_into_node(s, first + slot, cc, ch, depth + 1, type, index);
        }
    } else {
        node_push_rec(&s->nodes[ni], type, index);
    }
}

static void *dup_mem(const void *p, size_t n)
{
    void *q = malloc(n ? n : 1);
    if (n) memcpy(q, p, n);
    return q;
}

o_scene *oracle_scene_create(const o_scene_desc *d)
{
    o_scene *s = (o_scene *)calloc(1, sizeof(o_scene));
    uint32_t i, k, nt = 0;
    o_v3 root_c, root_h;
    s->materials = (o_material *)dup_mem(d->materials, sizeof(o_material) * d->material_count);
    s->material_count = d->material_count;
    s->spheres = (o_sphere *)dup_mem(d->spheres, sizeof(o_sphere) * d->sphere_count);
    s->sphere_count = d->sphere_count;
    s->boxes = (o_box *)dup_mem(d->boxes, sizeof(o_box) * d->box_count);
    s->box_count = d->box_count;
    s->cylinders = (o_cylinder *)dup_mem(d->cylinders, sizeof(o_cylinder) * d->cylinder_count);
    s->cylinder_count = d->cylinder_count;
    s->lights = (o_light *)dup_mem(d->lights, sizeof(o_light) * d->light_count);
    s->light_count = d->light_count;
    s->meshes = (o_mesh *)dup_mem(d->meshes, sizeof(o_mesh) * d->mesh_count);
    s->mesh_count = d->mesh_count;
    for (i = 0; i < d->mesh_count; ++i) {
        s->meshes[i].vertices = (const float *)dup_mem(d->meshes[i].vertices, 12 * (size_t)d->meshes[i].vertex_count);
        s->meshes[i].indices = (const uint32_t *)dup_mem(d->meshes[i].indices, 4 * (size_t)d->meshes[i].index_count);
        nt += d->meshes[i].index_count / 3;
    }
    s->tris = (o_tri *)malloc(sizeof(o_tri) * (nt ? nt : 1));
    for (i = 0; i < d->mesh_count; ++i)
        for (k = 0; k + 2 < d->meshes[i].index_count; k += 3) {
            o_tri *t = &s->tris[s->tri_count++];
            t->mesh = i;
            t->i0 = s->meshes[i].indices[k];
            t->i1 = s->meshes[i].indices[k + 1];
            t->i2 = s->meshes[i].indices[k + 2];
        }
    s->depth_limit = d->octree_depth ? d->octree_depth : 10;

    /* root node + root AABB in main()'s order (macos_main.mm:418-472) */
    s->node_cap = 1024;
    s->nodes = (o_node *)malloc(sizeof(o_node) * s->node_cap);
    memset(&s->nodes[0], 0, sizeof(o_node)); /* zero(top_most_node): is_leaf = 0 */
    s->nodes[0].first_child = -1;
    s->nodes[0].lo = v3(FLT_MAX, FLT_MAX, FLT_MAX);
    s->nodes[0].hi = v3(FLT_MIN, FLT_MIN, FLT_MIN);
    s->node_count = 1;
    for (i = 0; i < s->mesh_count; ++i) grow_aabb(s, &s->nodes[0].lo, &s->nodes[0].hi, T_MESH, i);
    for (i = 0; i < s->cylinder_count; ++i) grow_aabb(s, &s->nodes[0].lo, &s->nodes[0].hi, T_CYL, i);
    for (i = 0; i < s->box_count; ++i) grow_aabb(s, &s->nodes[0].lo, &s->nodes[0].hi, T_AAB, i);
    for (i = 0; i < s->sphere_count; ++i) grow_aabb(s, &s->nodes[0].lo, &s->nodes[0].hi, T_SPHERE, i);
    if (d->with_reference_csg) {
        /* macos_main.mm:322-332,462-469: sphere r=0.35 and box +-0.3 around (0,0,0.8) */
        o_v3 c = v3(0, 0, 0.8f);
        o_v3 lo = v_scale(FLT_MAX, v3(1, 1, 1)), hi = v_scale(FLT_MIN, v3(1, 1, 1));
        o_v3 sc = c, sh = v_scale(0.35f, v3(1, 1, 1));
        o_v3 bl = v_sub(c, v3(0.3f, 0.3f, 0.3f)), bh = v_add(c, v3(0.3f, 0.3f, 0.3f));
        o_v3 bc = v_scale(0.5f, v_add(bl, bh)), bhd = v_sub(bh, bc);
        lo = v_min(lo, v_sub(sc, sh)); hi = v_max(hi, v_add(sc, sh));
        lo = v_min(lo, v_sub(bc, bhd)); hi = v_max(hi, v_add(bc, bhd));
        s->has_csg = 1; s->csg_lo = lo; s->csg_hi = hi;
    }
    root_c = v_scale(0.5f, v_add(s->nodes[0].lo, s->nodes[0].hi));
    root_h = v_sub(s->nodes[0].hi, root_c);
    /* pushes (macos_main.mm:478-538): triangles, cylinders, boxes, spheres, CSG */
    for (i = 0; i < s->tri_count; ++i) push_into_node(s, 0, root_c, root_h, 0, T_TRI, i);
    for (i = 0; i < s->cylinder_count; ++i) push_into_node(s, 0, root_c, root_h, 0, T_CYL, i);
    for (i = 0; i < s->box_count; ++i) push_into_node(s, 0, root_c, root_h, 0, T_AAB, i);
    for (i = 0; i < s->sphere_count; ++i) push_into_node(s, 0, root_c, root_h, 0, T_SPHERE, i);
    if (s->has_csg) push_into_node(s, 0, root_c, root_h, 0, T_CSG, 0);
    return s;
}

void oracle_scene_destroy(o_scene *s)
{
    uint32_t i;
    if (!s) return;
    for (i = 0; i < s->node_count; ++i) free(s->nodes[i].recs);
    for (i = 0; i < s->mesh_count; ++i) { free((void *)s->meshes[i].vertices); free((void *)s->meshes[i].indices); }
    free(s->nodes); free(s->tris); free(s->meshes); free(s->lights); free(s->cylinders);
    free(s->boxes); free(s->spheres); free(s->materials); free(s);
}

void oracle_tree_stats(const o_scene *s, o_tree_stats *out)
{
    uint32_t i, k;
    memset(out, 0, sizeof(*out));
    out->nodes = s->node_count;
    for (i = 0; i < s->node_count; ++i) {
        uint64_t b = 0;
        if (!s->nodes[i].nrecs) continue;
        out->nonempty_leaves++;
        if (s->nodes[i].nrecs > out->max_leaf_records) out->max_leaf_records = s->nodes[i].nrecs;
        for (k = 0; k < s->nodes[i].nrecs; ++k) b += k_rec_bytes[s->nodes[i].recs[k].type];
        out->record_bytes += b;
    }
}

/* ---- raycast (ray.cpp:603-822,1165-1176) -------------------------------------------- */
typedef struct { uint32_t *q; uint32_t cap; } o_queue;
typedef struct { float t; o_v3 n; uint32_t mat; uint32_t tested; int32_t inner; } o_cast;

static void raycast(const o_scene *s, o_queue *Q, o_v3 o, o_v3 d, o_cast *res, o_render_stats *st)
{
    uint32_t used = 0, rd = 0;
    memset(res, 0, sizeof(*res));
    res->t = FLT_MAX;
    Q->q[used++] = 0; /* raycast_top_most_node pushes the root, ray.cpp:1170 */
    while (rd < used) {
        const o_node *node = &s->nodes[Q->q[rd]];
        uint32_t k;
        if (st) st->node_pops++;
        for (k = 0; k < node->nrecs; ++k) {
            o_rec rec = node->recs[k];
            o_hit h;
            uint32_t mat;
            switch (rec.type) {
            case T_SPHERE: {
                const o_sphere *sp = &s->spheres[rec.index];
                h = hit_sphere(sp->center, sp->r, o, d); mat = sp->mat;
                if (st) st->analytic_tests++;
            } break;
            case T_AAB: {
                const o_box *b = &s->boxes[rec.index];
                h = hit_aab(b->min, b->max, o, d); mat = b->mat;
                if (st) st->analytic_tests++;
            } break;
            case T_CYL: {
                const o_cylinder *c = &s->cylinders[rec.index];
                h = hit_cylinder(c->base, c->axis, c->r, o, d); mat = c->mat;
                if (st) st->analytic_tests++;
            } break;
            case T_TRI: {
                const o_tri *t = &s->tris[rec.index];
                h = hit_triangle(tri_vertex(s, t->mesh, t->i0), tri_vertex(s, t->mesh, t->i1), tri_vertex(s, t->mesh, t->i2), o, d);
                mat = s->meshes[t->mesh].mat;
                if (st) st->tri_tests++;
            } break;
            default: continue; /* CSG: hit test compiled out, not counted (ray.cpp:718-767) */
            }
            if (h.t >= O_HIT_T_MIN && h.t < res->t) {
                res->t = h.t; res->n = h.n; res->mat = mat; res->inner = h.inner;
            }
            res->tested++;
        }
        if (node->first_child >= 0) {
            uint32_t c;
            for (c = 0; c < 8; ++c) {
                uint32_t ci = (uint32_t)node->first_child + c;
                const o_node *ch = &s->nodes[ci];
                if ((ch->is_leaf && ch->nrecs) || ch->first_child >= 0) {
                    int add = 0;
                    if (v_in_rect(o, ch->lo, ch->hi)) {
                        add = 1;
                    } else {
                        o_hit h = hit_aab(ch->lo, ch->hi, o, d);
                        if (st) st->child_tests++;
                        if (h.t >= O_HIT_T_MIN && h.t < res->t) add = 1;
                    }
                    if (add) {
                        if (used == Q->cap) {
                            Q->cap *= 2;
                            Q->q = (uint32_t *)realloc(Q->q, sizeof(uint32_t) * Q->cap);
                        }
                        Q->q[used++] = ci;
                    }
                }
            }
        }
        rd++;
    }
    res->n = v_normalize(res->n);
    if (st) { st->rays++; st->shapes_tested += res->tested; }
}

void oracle_raycast(const o_scene *s, const float origin[3], const float dir[3], float *t, float normal[3], uint32_t *mat)
{
    o_queue Q;
    o_cast r;
    Q.cap = 4096;
    Q.q = (uint32_t *)malloc(sizeof(uint32_t) * Q.cap);
    raycast(s, &Q, v3(origin[0], origin[1], origin[2]), v3(dir[0], dir[1], dir[2]), &r, 0);
    *t = r.t; normal[0] = r.n.x; normal[1] = r.n.y; normal[2] = r.n.z; *mat = r.mat;
    free(Q.q);
}

/* ray.cpp:537-601: the sampled point is dead (its only consumer is #if 0, ray.cpp:1285-1327)
   but the RNG advances: 1 for the index, +4 when the chosen entry is a sphere. */
static void burn_light_sample(const o_scene *s, uint32_t *rng)
{
    uint32_t idx;
    if (s->light_count == 0) { rng_step(rng); return; } /* reference: % 0 (UB); defined here */
    idx = rng_between_u32(rng, 0, s->light_count);
    if (s->lights[idx].type == T_SPHERE) {
        (void)rng_spherical(rng, -O_PI / 2.0f, O_PI / 2.0f, 0, 2.0f * O_PI);
    }
}

uint64_t oracle_tiled_raytrace(const o_scene *s, const o_camera *cam, float *out, int32_t W, int32_t H, int32_t x0,
                               int32_t y0, int32_t x1, int32_t y1, uint32_t *rng, uint32_t spp, float rr,
                               o_render_stats *st) /* ray.cpp:1178-1466 */
{
    uint64_t tested = 0;
    o_queue Q;
    float roughness = 0.01f, eps = 0.0001f;
    float focal_length = v_len(v_sub(cam->p, v3(0, 0, 0.2f)));
    float aperture_radius = 0.1f;
    int32_t x, y;
    Q.cap = 4096;
    Q.q = (uint32_t *)malloc(sizeof(uint32_t) * Q.cap);
    for (y = y0; y < y1; ++y) {
        for (x = x0; x < x1; ++x) {
            o_v3 color = v3(0, 0, 0);
            float px = (2.0f * x / (float)W) - 1.0f;
            float py = (2.0f * y / (float)H) - 1.0f;
            o_v3 to_pixel = v_normalize(v_sub(v_add(v_scale(px, cam->x_axis), v_scale(py, cam->y_axis)), cam->z_axis));
            o_v3 focal = v_add(cam->p, v_scale(focal_length, to_pixel));
            uint32_t si;
            for (si = 0; si < spp; ++si) {
                float rad = rng_between(rng, 0.0f, 2 * O_PI);
                o_v3 ap = v_sub(v_add(v_add(cam->p, v_scale(aperture_radius * det_cosf(rad), cam->x_axis)),
                                      v_scale(aperture_radius * det_sinf(rad), cam->y_axis)),
                                v_scale(0.1f, cam->z_axis));
                o_v3 dir0 = v_normalize(v_sub(focal, ap));
                o_v3 wo = v_neg(v_normalize(dir0));
                o_v3 origin = ap, prev_dir = v3(0, 0, 0), hit_n = v3(0, 0, 0), weight = v3(1, 1, 1);
                const o_material *hit_mat = 0;
                int alive = 1;
                o_cast r0;
                raycast(s, &Q, origin, dir0, &r0, st);
                tested += r0.tested;
                if (st) st->paths++;
                if (r0.mat) {
                    const o_material *m = &s->materials[r0.mat];
                    if (m->is_light) {
                        color = v_add(color, m->emit);
                        alive = 0;
                    } else {
                        origin = v_add(origin, v_scale(r0.t - eps, dir0));
                        hit_n = r0.n;
                        hit_mat = m;
                        prev_dir = dir0;
                        if (v_len2(m->diffuse) > 0.0f) weight = v_had(weight, m->diffuse);
                    }
                } else {
                    alive = 0; /* reference: undefined behaviour on a primary miss; defined: terminate */
                }
                while (alive && rng_01(rng) < rr) {
                    o_v3 wi;
                    int is_trans;
                    o_cast r;
                    burn_light_sample(s, rng);
                    wi = bsdf_sample(rng, hit_n, wo, roughness, hit_mat->diffuse,
                                     v3(hit_mat->specular[0], hit_mat->specular[1], hit_mat->specular[2]),
                                     hit_mat->transmission, hit_mat->ior, &is_trans);
                    if (is_trans) origin = v_add(origin, v_scale(2.0f * eps, prev_dir));
                    raycast(s, &Q, origin, wi, &r, st);
                    tested += r.tested;
                    if (r.mat) {
                        const o_material *m = &s->materials[r.mat];
                        if (m->is_light) {
                            o_v3 c = v_had(weight, m->emit);
                            if (!v_isnan(c) && !v_isinf(c)) color = v_add(color, c);
                            alive = 0;
                        } else {
                            o_v3 ks = v3(m->specular[0], m->specular[1], m->specular[2]);
                            float p = bsdf_pdf(r.n, wi, wo, roughness, m->diffuse, ks, m->transmission, m->ior) * rr;
                            if (p > 0.000001f) {
                                o_v3 f = bsdf_eval(r.n, wi, wo, m->diffuse, ks, m->transmission, m->ior, roughness, r.t);
                                weight = v_had(v_div(f, p), weight);
                            }
                            origin = v_add(origin, v_scale(r.t - eps, wi));
                            hit_n = r.n;
                            hit_mat = m;
                            prev_dir = wi;
                            wo = v_neg(wi);
                        }
                    } else {
                        alive = 0;
                    }
                }
            }
            {
                o_v3 px_out = v_div(color, (float)spp);
                float *dst = out + 3 * ((size_t)y * (size_t)W + (size_t)x);
                dst[0] = px_out.x; dst[1] = px_out.y; dst[2] = px_out.z;
            }
        }
    }
    free(Q.q);
    if (st) st->final_rng = *rng;
    return tested;
}

/* ---- caller policies, multithreaded (one job = one reference call) ------------------- */
typedef struct {
    const o_scene *s; const o_camera *cam; float *out; int32_t W, H, x0, y0, x1, y1;
    int32_t policy; uint32_t seed, spp, chunk; float rr;
    uint32_t *tile_seeds; int32_t tw, th;
    volatile int64_t next; int64_t njobs;
    pthread_mutex_t mu;
    o_render_stats total;
} o_work;

static void stats_add(o_render_stats *a, const o_render_stats *b)
{
    a->paths += b->paths; a->rays += b->rays; a->node_pops += b->node_pops; a->child_tests += b->child_tests;
    a->tri_tests += b->tri_tests; a->analytic_tests += b->analytic_tests; a->shapes_tested += b->shapes_tested;
}

static void run_job(o_work *w, int64_t j, o_render_stats *st)
{
    if (w->policy == O_POLICY_TILE32) {
        int32_t tx = (int32_t)(j % 32), ty = (int32_t)(j / 32);
        int32_t ax = tx * w->tw, ay = ty * w->th, bx = ax + w->tw, by = ay + w->th;
        uint32_t rng = w->tile_seeds[j];
        if (bx > w->W) bx = w->W;
        if (by > w->H) by = w->H;
        if (ax < w->x0 || ay < w->y0 || bx > w->x1 || by > w->y1) return; /* whole tiles inside the rect only */
        oracle_tiled_raytrace(w->s, w->cam, w->out, w->W, w->H, ax, ay, bx, by, &rng, w->spp, w->rr, st);
    } else if (w->policy == O_POLICY_WHOLE) {
        uint32_t rng = w->tile_seeds[0];
        oracle_tiled_raytrace(w->s, w->cam, w->out, w->W, w->H, w->x0, w->y0, w->x1, w->y1, &rng, w->spp, w->rr, st);
    } else {
        int32_t rw = w->x1 - w->x0;
        int32_t x = w->x0 + (int32_t)(j % rw), y = w->y0 + (int32_t)(j / rw);
        uint32_t pix = (uint32_t)(y * w->W + x);
        if (w->policy == O_POLICY_PIXEL) {
            uint32_t rng = oracle_job_seed(w->seed, pix);
            oracle_tiled_raytrace(w->s, w->cam, w->out, w->W, w->H, x, y, x + 1, y + 1, &rng, w->spp, w->rr, st);
        } else {
            uint32_t nch = w->spp / w->chunk, k;
            float *dst = w->out + 3 * (size_t)pix;
            o_v3 acc = v3(0, 0, 0);
            for (k = 0; k < nch; ++k) {
                uint32_t rng = oracle_job_seed(w->seed, k * (uint32_t)(w->W * w->H) + pix);
                oracle_tiled_raytrace(w->s, w->cam, w->out, w->W, w->H, x, y, x + 1, y + 1, &rng, w->chunk, w->rr, st);
                acc = v_add(acc, v3(dst[0], dst[1], dst[2]));
            }
            acc = v_div(acc, (float)nch);
            dst[0] = acc.x; dst[1] = acc.y; dst[2] = acc.z;
        }
    }
}

static void *worker(void *arg)
{
    o_work *w = (o_work *)arg;
    o_render_stats st;
    memset(&st, 0, sizeof(st));
    for (;;) {
        int64_t j = __sync_fetch_and_add(&w->next, 1);
        if (j >= w->njobs) break;
        run_job(w, j, &st);
    }
    pthread_mutex_lock(&w->mu);
    stats_add(&w->total, &st);
    w->total.final_rng = st.final_rng;
    pthread_mutex_unlock(&w->mu);
    return 0;
}

int oracle_render_image(const o_scene *s, const o_camera *cam, float *out, int32_t W, int32_t H, int32_t x0, int32_t y0,
                        int32_t x1, int32_t y1, int32_t policy, uint32_t seed, uint32_t spp, uint32_t chunk, float rr,
                        int32_t threads, o_render_stats *stats)
{
    o_work w;
    struct timespec t0, t1;
    pthread_t th[256];
    int32_t i;
    memset(&w, 0, sizeof(w));
    w.s = s; w.cam = cam; w.out = out; w.W = W; w.H = H; w.x0 = x0; w.y0 = y0; w.x1 = x1; w.y1 = y1;
    w.policy = policy; w.seed = seed; w.spp = spp; w.chunk = chunk; w.rr = rr;
    pthread_mutex_init(&w.mu, 0);
    if (threads < 1) threads = 1;
    if (threads > 256) threads = 256;
    if (policy == O_POLICY_TILE32) {
        uint32_t master = seed;
        w.tile_seeds = (uint32_t *)malloc(4 * 1024);
        for (i = 0; i < 1024; ++i) w.tile_seeds[i] = rng_u32(&master); /* macos_main.mm:644 */
        w.tw = (int32_t)ceilf(W / (float)32);                          /* macos_main.mm:604-605 */
        w.th = (int32_t)ceilf(H / (float)32);
        w.njobs = 1024;
    } else if (policy == O_POLICY_WHOLE) {
        uint32_t master = seed;
        w.tile_seeds = (uint32_t *)malloc(4);
        w.tile_seeds[0] = rng_u32(&master);
        w.njobs = 1;
    } else if (policy == O_POLICY_PIXEL || policy == O_POLICY_CHUNK) {
        if (policy == O_POLICY_CHUNK && (chunk == 0 || spp % chunk)) return 1;
        w.njobs = (int64_t)(x1 - x0) * (int64_t)(y1 - y0);
    } else {
        return 1;
    }
    clock_gettime(CLOCK_MONOTONIC, &t0);
    if (threads == 1) {
        worker(&w);
    } else {
        for (i = 0; i < threads; ++i) pthread_create(&th[i], 0, worker, &w);
        for (i = 0; i < threads; ++i) pthread_join(th[i], 0);
    }
    clock_gettime(CLOCK_MONOTONIC, &t1);
    w.total.seconds = (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
    if (stats) *stats = w.total;
    free(w.tile_seeds);
    pthread_mutex_destroy(&w.mu);
    return 0;
}

/* ---- per-function tables -------------------------------------------------------------- */
static o_v3 in3(const float *p) { return v3(p[0], p[1], p[2]); }
static void out_hit(float *o, o_hit h) { o[0] = h.t; o[1] = h.n.x; o[2] = h.n.y; o[3] = h.n.z; o[4] = (float)h.inner; }

void oracle_unit(uint32_t op, const float a[24], float o[8])
{
    memset(o, 0, 32);
    switch (op) {
    case 1: out_hit(o, hit_triangle(in3(a), in3(a + 3), in3(a + 6), in3(a + 9), in3(a + 12))); break;
    case 2: out_hit(o, hit_sphere(in3(a), a[3], in3(a + 4), in3(a + 7))); break;
    case 3: out_hit(o, hit_aab(in3(a), in3(a + 3), in3(a + 6), in3(a + 9))); break;
    case 4: out_hit(o, hit_cylinder(in3(a), in3(a + 3), a[6], in3(a + 7), in3(a + 10))); break;
    case 5: {
        uint32_t seed;
        int tr;
        o_v3 wi;
        memcpy(&seed, a, 4);
        wi = bsdf_sample(&seed, in3(a + 1), in3(a + 4), a[7], in3(a + 8), in3(a + 11), in3(a + 14), a[17], &tr);
        o[0] = wi.x; o[1] = wi.y; o[2] = wi.z; o[3] = (float)tr;
        memcpy(o + 4, &seed, 4);
    } break;
    case 6: o[0] = bsdf_pdf(in3(a), in3(a + 3), in3(a + 6), a[9], in3(a + 10), in3(a + 13), in3(a + 16), a[19]); break;
    case 7: {
        o_v3 f = bsdf_eval(in3(a), in3(a + 3), in3(a + 6), in3(a + 9), in3(a + 12), in3(a + 15), a[18], a[19], a[20]);
        o[0] = f.x; o[1] = f.y; o[2] = f.z;
    } break;
    case 8: {
        o_v3 r = bsdf_sample_lobe(in3(a), a[3], a[4]);
        o[0] = r.x; o[1] = r.y; o[2] = r.z;
    } break;
    case 9:
        o[0] = det_sinf(a[0]); o[1] = det_cosf(a[0]); o[2] = det_atan2f(a[1], a[0]);
        o[3] = det_powf(a[0], a[1]); o[4] = det_logf(a[0]);
        break;
    case 10: {
        o_v3 r = v_normalize(in3(a));
        o[0] = r.x; o[1] = r.y; o[2] = r.z;
    } break;
    case 11: {
        o_v3 F = bsdf_fresnel(in3(a), a[3]);
        o[0] = F.x; o[1] = F.y; o[2] = F.z;
        o[3] = bsdf_ggx(in3(a + 4), in3(a + 7), a[10]);
        o[4] = bsdf_geometry(in3(a + 11), in3(a + 4), in3(a + 7), a[10]);
    } break;
    default: break;
    }
}

void oracle_unit_batch(const uint8_t *records, uint64_t n, float *out)
{
    uint64_t i;
    for (i = 0; i < n; ++i) {
        uint32_t op;
        float a[24];
        memcpy(&op, records + i * 100, 4);
        memcpy(a, records + i * 100 + 4, 96);
        oracle_unit(op, a, out + 8 * i);
    }
}

void oracle_rng_table(uint32_t seed, uint32_t n, uint8_t *out)
{
    uint32_t i, s;
    uint8_t *p = out;
    s = seed;
    for (i = 0; i < n; ++i) { float v = rng_01(&s); memcpy(p, &s, 4); memcpy(p + 4, &v, 4); p += 8; }
    s = seed;
    for (i = 0; i < n; ++i) { float v = rng_between(&s, 0.0f, 2 * O_PI); memcpy(p, &v, 4); p += 4; }
    s = seed;
    for (i = 0; i < n; ++i) { uint32_t v = rng_between_u32(&s, 0, 12); memcpy(p, &v, 4); p += 4; }
    s = seed;
    for (i = 0; i < n; ++i) { o_v3 v = rng_spherical(&s, -O_PI / 2.0f, O_PI / 2.0f, 0, 2.0f * O_PI); memcpy(p, &v, 12); p += 12; }
    memcpy(p, &s, 4);
}

/* ---- output (macos_main.mm:242-287,683-707) -------------------------------------------- */
uint32_t oracle_rgbe(float r, float g, float b)
{
    uint32_t result = 0;
    float mx = O_MAX(O_MAX(r, g), b);
    int e;
    if (mx >= 1e-32f) {
        float denom = frexpf(mx, &e) * 255.0f / mx; /* C++ picks the float overload of frexp */
        result = (((uint32_t)roundf(r * denom) << 0) | ((uint32_t)roundf(g * denom) << 8) |
                  ((uint32_t)roundf(b * denom) << 16) | ((uint32_t)(e + 128) << 24));
    }
    return result;
}

int oracle_write_hdr(const char *path, const float *rgb, int32_t W, int32_t H)
{
    FILE *f = fopen(path, "wb");
    int32_t x, y;
    if (!f) return 1;
    fprintf(f, "#?RADIANCE\n");
    fprintf(f, "FORMAT=32-bit_rle_rgbe\n\n");
    fprintf(f, "+Y %d +X %d\n", H, W);
    for (y = H - 1; y >= 0; --y)
        for (x = 0; x < W; ++x) {
            const float *p = rgb + 3 * ((size_t)y * W + x);
            uint32_t c = oracle_rgbe(p[0], p[1], p[2]);
            fwrite(&c, 4, 1, f);
        }
    fclose(f);
    return 0;
}

/* quaternion rotation, math.h:771-793 */
static o_v3 quat_rotate(const float q[4], o_v3 v)
{
    float x = q[0], y = q[1], z = q[2], w = q[3];
    float m00 = 1.0f - 2 * y * y - 2 * z * z, m01 = 2 * x * y - 2 * w * z, m02 = 2 * x * z + 2 * w * y;
    float m10 = 2 * x * y + 2 * w * z, m11 = 1.0f - 2 * x * x - 2 * z * z, m12 = 2 * y * z - 2 * w * x;
    float m20 = 2 * x * z - 2 * w * y, m21 = 2 * y * z + 2 * w * x, m22 = 1 - 2 * x * x - 2 * y * y;
    return v3(m00 * v.x + m01 * v.y + m02 * v.z, m10 * v.x + m11 * v.y + m12 * v.z, m20 * v.x + m21 * v.y + m22 * v.z);
}

void oracle_camera(const float p[3], const float q[4], float ratio, int32_t W, int32_t H, o_camera *out) /* macos_main.mm:550-556 */
{
    float rx = ratio * ((float)W / H);
    out->p = v3(p[0], p[1], p[2]);
    out->x_axis = v_scale(rx, quat_rotate(q, v3(1, 0, 0)));
    out->y_axis = v_scale(ratio, quat_rotate(q, v3(0, 1, 0)));
    out->z_axis = quat_rotate(q, v3(0, 0, 1));
}
