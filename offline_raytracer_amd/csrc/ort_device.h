/*
 * ort_device.h -- gfx950 device functions of the path-trace hot path.
 *
 * What each function must reproduce is the reference's scalar f32 arithmetic, operation
 * for operation (paths relative to /root/reference/code); built with -ffp-contract=off,
 * IEEE divide/sqrt (hipcc default), denormals on.  "sic" marks reference oddities that are
 * part of its results.
 */
#ifndef ORT_DEVICE_H
#define ORT_DEVICE_H

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ort_detmath.h"

namespace ortd {

#ifdef ORT_HOST_SIM /* developer harness only (tools/host_sim.cpp); never defined in a product build */
#define ORT_D static inline
#else
#define ORT_D __device__ __forceinline__
#endif

struct V3 { float x, y, z; };

constexpr float kPi = 3.14159265358979323846264338327950288419716939937510582097494459230f; /* platform.h:44 */
constexpr float kEuler = 2.71828182845904523536028747135266249f;                            /* ray.cpp:4 */
constexpr float kHitTMin = 0.000001f;                                                       /* ray.cpp:5 */
constexpr float kRoughness = 0.01f;                                                         /* ray.cpp:1194 */
constexpr float kEps = 0.0001f;                                                             /* ray.cpp:1196 */

ORT_D V3 mk(float x, float y, float z) { V3 r; r.x = x; r.y = y; r.z = z; return r; }
ORT_D V3 add(V3 a, V3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
ORT_D V3 sub(V3 a, V3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
ORT_D V3 neg(V3 a) { return mk(-a.x, -a.y, -a.z); }
ORT_D V3 scale(float s, V3 a) { return mk(s * a.x, s * a.y, s * a.z); }
ORT_D V3 divs(V3 a, float s) { return mk(a.x / s, a.y / s, a.z / s); }            /* three divides, math.h:234-244 */
ORT_D V3 had(V3 a, V3 b) { return mk(a.x * b.x, a.y * b.y, a.z * b.z); }
ORT_D float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }        /* math.h:319-323 */
ORT_D float len2(V3 a) { return a.x * a.x + a.y * a.y + a.z * a.z; }
ORT_D float len(V3 a) { return __builtin_sqrtf(a.x * a.x + a.y * a.y + a.z * a.z); }
ORT_D V3 cross(V3 a, V3 b) {                                                      /* math.h:280-290 */
    return mk(a.y * b.z - b.y * a.z, b.x * a.z - a.x * b.z, a.x * b.y - b.x * a.y);
}
ORT_D float absr(float v) { if (v <= 0.0f) v *= -1.0f; return v; }                /* intrinsic.h:132-143 (+0 -> -0) */
ORT_D float sq(float v) { return v * v; }
ORT_D bool ceq(float a, float b) { float d = a - b; return d >= -0.000001f && d < 0.000001f; } /* math.h:9-22 */
ORT_D float sgn(float a) { return (a >= 0.0f) ? 1.0f : -1.0f; }                   /* types.h:52 */
ORT_D float rmin(float a, float b) { return (a < b) ? a : b; }                    /* types.h:51: NaN -> b */
ORT_D float rmax(float a, float b) { return (a > b) ? a : b; }                    /* types.h:50: NaN -> b */
ORT_D V3 normalize(V3 a) {                                                        /* math.h:298-310 */
    float l = len(a);
    if (__builtin_expect(!ceq(l, 0.0f), 1)) return divs(a, l);
    return mk(0, 0, 0);
}
ORT_D bool isnan3(V3 v) { return (v.x != v.x) || (v.y != v.y) || (v.z != v.z); }
ORT_D bool isinf1(float v) { return (om_f32_bits(v) & 0x7fffffffu) == 0x7f800000u; }
ORT_D bool isinf3(V3 v) { return isinf1(v.x) || isinf1(v.y) || isinf1(v.z); }

/* ---- RNG: random.h:5-117 --------------------------------------------------------------- */
ORT_D void rng_step(uint32_t &s) { s ^= s << 13; s ^= s >> 17; s ^= s >> 5; }     /* sic: third shift is right */
ORT_D float rng_01(uint32_t &s) { rng_step(s); return (float)s / 4294967296.0f; } /* (f32)U32_Max == 2^32 */
ORT_D float rng_between(uint32_t &s, float lo, float hi) { rng_step(s); return lo + (hi - lo) * rng_01(s); } /* sic: two steps */

ORT_D uint32_t fmix32(uint32_t h) {
    h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16;
    return h;
}
ORT_D uint32_t job_seed(uint32_t master, uint32_t job) {
    uint32_t h = fmix32(master ^ (job * 2654435761u));
    return h ? h : 1u;
}

/* ---- material record (DevMaterial, 5 x float4) ------------------------------------------- */
struct Mat {
    V3 kd; float ior;
    V3 ks; uint32_t is_light;
    V3 kt; float pd_c;  /* |Kd| / (|Kd|+|Ks|+|Kt|), ray.cpp:1016,1111 */
    V3 emit; float ps_c;
    V3 ed; float pt_c;  /* Kd / pi, ray.cpp:939 */
};
ORT_D Mat load_mat(const float4 *mats, uint32_t index) {
    const float4 *p = mats + 5u * index;
    float4 a = p[0], b = p[1], c = p[2], d = p[3], e = p[4];
    Mat m;
    m.kd = mk(a.x, a.y, a.z); m.ior = a.w;
    m.ks = mk(b.x, b.y, b.z); m.is_light = om_f32_bits(b.w);
    m.kt = mk(c.x, c.y, c.z); m.pd_c = c.w;
    m.emit = mk(d.x, d.y, d.z); m.ps_c = d.w;
    m.ed = mk(e.x, e.y, e.z); m.pt_c = e.w;
    return m;
}
/* the same record from raw coefficients (unit tests): the per-material values as the reference computes them */
ORT_D Mat make_mat(V3 kd, V3 ks, V3 kt, float ior) {
    Mat m;
    m.kd = kd; m.ks = ks; m.kt = kt; m.ior = ior; m.is_light = 0; m.emit = mk(0, 0, 0);
    float a = len(kd), b = len(ks), c = len(kt);
    float s = a + b + c;
    m.pd_c = a / s; m.ps_c = b / s; m.pt_c = c / s;
    m.ed = divs(kd, kPi);
    return m;
}

/* ---- BSDF: ray.cpp:825-1161 ------------------------------------------------------------- */
ORT_D V3 fresnel(V3 ks, float l_dot_h) {                                          /* ray.cpp:825-831, sic */
    float k = 1 - ort_powf(1.0f - absr(l_dot_h), 5.0f);
    return add(ks, scale(k, sub(mk(1, 1, 1), ks)));
}
ORT_D float ggx_d(V3 N, V3 H, float rough) {                                      /* ray.cpp:834-865 */
    float result = 0.0f, ndh = dot(N, H);
    if (ndh > 0.0f) {
        float r2 = sq(rough);
        float tan_t = __builtin_sqrtf(1.0f - sq(ndh)) / ndh;
        float denom = kPi * ort_powf(ndh, 4.0f) * sq(r2 + sq(tan_t));
        if (!ceq(denom, 0.0f)) result = r2 / denom;
    }
    return result;
}
ORT_D float geom(V3 w, V3 N, V3 m, float rough) {                                 /* ray.cpp:868-897 */
    float result = 0.0f, wdn = dot(w, N), wdm = dot(w, m);
    if (!ceq(wdm, 0.0f) && (wdn / wdm) > 0) {
        if (wdm > 1.0f) {
            result = 1.0f;
        } else {
            float tan_t = __builtin_sqrtf(1.0f - sq(wdn)) / wdn;
            if (!ceq(tan_t, 0.0f)) {
                float r2 = sq(rough);
                result = 2.0f / (1.0f + __builtin_sqrtf(1 + r2 * sq(tan_t)));
            }
        }
    }
    return result;
}
ORT_D float radicand(V3 m, V3 wo, float n) { return 1 - sq(n) * (1 - sq(dot(wo, m))); } /* ray.cpp:899-904 */
struct Beer { float ni, no, n; };
ORT_D Beer beer(V3 N, V3 wo, float ior) {                                         /* ray.cpp:914-933 */
    Beer r;
    if (dot(N, wo) >= 0.0f) { r.ni = 1.0f; r.no = ior; } else { r.ni = ior; r.no = 1.0f; }
    r.n = r.ni / r.no;
    return r;
}

/* DIFFUSE_ONLY = compiled for a scene in which no material can enter the specular or the
   transmission block (|Ks|^2 > 0, |Kt|^2 > 0, ps_c > 0, pt_c > 0 all false; checked on the host at
   upload): those blocks are then dead code and dropping them frees registers.  Values are unchanged. */
template <bool DIFFUSE_ONLY = false>
ORT_D V3 eval_scattering(V3 N, V3 wi, V3 wo, const Mat &mt, float rough, float dist) { /* ray.cpp:936-1005 */
    V3 Ed = mt.ed; /* Kd / pi_32, per material */
    V3 Es = mk(0, 0, 0), Et = mk(0, 0, 0);
    float wi_n = dot(wi, N), wo_n = dot(wo, N);
    if (DIFFUSE_ONLY) return scale(absr(wi_n), add(add(Ed, Es), Et));
    /* H is only consumed under "wi.H > 0 && |Ks|^2 > 0" (ray.cpp:949): not computed for Ks = 0 */
    if (len2(mt.ks) > 0.0f) {
        V3 H = scale(sgn(dot(wi, N)), normalize(add(wo, wi)));
        float wi_h = dot(wi, H);
        if (wi_h > 0.0f) {
        V3 F = fresnel(mt.ks, wi_h);
        float D = ggx_d(N, H, rough);
        float G = geom(wi, N, H, rough) * geom(wo, N, H, rough);
        Es = scale((D * G) / (4.0f * absr(wi_n) * absr(wo_n)), F);
        }
    }
    if (len2(mt.kt) > 0.0f) {
        V3 At = mk(1, 1, 1);
        if (wo_n < 0) {                                                           /* sic: logf(0) = -inf is relied on */
            At.x = ort_powf(kEuler, dist * ort_logf(mt.kt.x));
            At.y = ort_powf(kEuler, dist * ort_logf(mt.kt.y));
            At.z = ort_powf(kEuler, dist * ort_logf(mt.kt.z));
        }
        Beer bn = beer(N, wo, mt.ior);
        V3 m = normalize(neg(add(scale(bn.ni, wi), scale(bn.no, wo))));
        float r = radicand(m, wo, bn.n);
        if (r < 0.0f) {
            if (len2(mt.ks) > 0.0f) Et = had(At, Es);
        } else {
            float wi_m = dot(wi, m), wo_m = dot(wo, m);
            V3 F = sub(mk(1, 1, 1), fresnel(mt.ks, wi_m));
            float D = ggx_d(N, m, rough);
            float G = geom(wi, N, m, rough) * geom(wo, N, m, rough);
            float denom = (absr(wi_n) * absr(wo_n) * sq(bn.ni * wi_m + bn.no * wo_m));
            if (!ceq(denom, 0.0f)) {
                V3 nom = scale(D * G * absr(wi_m) * absr(wo_m) * sq(bn.no), F);
                Et = had(At, divs(nom, denom));
            }
        }
    }
    return scale(absr(wi_n), add(add(Ed, Es), Et));
}

template <bool DIFFUSE_ONLY = false>
ORT_D float pdf_brdf(V3 N, V3 wi, V3 wo, float rough, const Mat &mt) {            /* ray.cpp:1007-1063 */
    float pd_c = mt.pd_c, ps_c = mt.ps_c, pt_c = mt.pt_c; /* per material, ray.cpp:1010-1018 */
    float pd = absr(dot(wi, N)) / kPi;
    float ps = 0.0f;
    if (!DIFFUSE_ONLY && ps_c > 0.0f) { /* H and its dot products are only consumed here */
        V3 H = scale(sgn(dot(N, wi)), normalize(add(wo, wi)));
        float n_h = dot(N, H), wi_h = dot(wi, H);
        float denom = (4.0f * absr(wi_h));
        if (!ceq(denom, 0.0f)) {
            float D = ggx_d(N, H, rough);
            ps = D * absr(n_h) / denom;
        }
    }
    float pt = ps;                                                                /* sic */
    if (!DIFFUSE_ONLY && pt_c > 0.0f) { /* m and the radicand are only consumed under "pt_c > 0 && r >= 0" */
        Beer bn = beer(N, wo, mt.ior);
        V3 m = normalize(neg(add(scale(bn.ni, wi), scale(bn.no, wo))));
        float r = radicand(m, wo, bn.n);
        if (r >= 0.0f) {
            float n_m = dot(N, m), wi_m = dot(wi, m), wo_m = dot(wo, m);
            float denom = sq(bn.no * wo_m + bn.no * wo_m);                        /* sic, ray.cpp:1054 */
            if (!ceq(denom, 0.0f)) {
                float D = ggx_d(N, m, rough);
                pt = D * absr(n_m) * sq(bn.no) * absr(wi_m) / denom;
            }
        }
    }
    return pd_c * pd + ps_c * ps + pt_c * pt;
}

/* pdf_brdf followed by eval_scattering on the same (N, wi, wo, material), as the path tracer calls them
   (ray.cpp:1374-1405): p = pdf * rr, and f only when p > 1e-6.  The two functions build the same half vector H, the
   same refraction normal m and the same GGX terms D(N, H), D(N, m) (the only differences are dot(N, wi) against
   dot(wi, N): the same products, summed in the same order); here each is evaluated once -- a power function in
   binary64 per D -- and consumed by both.  Every consumer keeps its own guard, so the values are those of the two
   separate calls (which the unit tables on the device keep testing).  Returns p; f is written when p > 1e-6. */
ORT_D float pdf_eval_scattering(V3 N, V3 wi, V3 wo, const Mat &mt, float rough, float dist, float rr, V3 &f) {
    const float wi_n = dot(wi, N), wo_n = dot(wo, N);
    const bool has_ks = len2(mt.ks) > 0.0f, has_kt = len2(mt.kt) > 0.0f;
    /* specular lobe: shared H, wi.H, D */
    V3 H = mk(0, 0, 0);
    float wi_h = 0.0f, D_h = 0.0f, ps = 0.0f;
    if (mt.ps_c > 0.0f || has_ks) {
        H = scale(sgn(wi_n), normalize(add(wo, wi)));
        wi_h = dot(wi, H);
        const float denom = (4.0f * absr(wi_h));
        const bool pdf_needs = mt.ps_c > 0.0f && !ceq(denom, 0.0f);
        if (pdf_needs || (has_ks && wi_h > 0.0f)) D_h = ggx_d(N, H, rough);
        if (pdf_needs) ps = D_h * absr(dot(N, H)) / denom;
    }
    /* transmission lobe: shared Beer indices, m, radicand, D */
    Beer bn; bn.ni = bn.no = bn.n = 0.0f;
    V3 m = mk(0, 0, 0);
    float r = -1.0f, D_m = 0.0f, wi_m = 0.0f, wo_m = 0.0f, pt = ps;                /* sic: pt starts as ps */
    if (mt.pt_c > 0.0f || has_kt) {
        bn = beer(N, wo, mt.ior);
        m = normalize(neg(add(scale(bn.ni, wi), scale(bn.no, wo))));
        r = radicand(m, wo, bn.n);
        if (!(r < 0.0f)) { /* r >= 0, or NaN: eval_scattering's else branch takes that too */
            wi_m = dot(wi, m); wo_m = dot(wo, m);
            D_m = ggx_d(N, m, rough);
        }
        if (r >= 0.0f && mt.pt_c > 0.0f) {
            const float denom = sq(bn.no * wo_m + bn.no * wo_m);                      /* sic, ray.cpp:1054 */
            if (!ceq(denom, 0.0f)) pt = D_m * absr(dot(N, m)) * sq(bn.no) * absr(wi_m) / denom;
        }
    }
    const float pd = absr(wi_n) / kPi;
    const float p = (mt.pd_c * pd + mt.ps_c * ps + mt.pt_c * pt) * rr;
    if (!(p > 0.000001f)) return p;
    /* eval_scattering, ray.cpp:936-1005 */
    V3 Es = mk(0, 0, 0), Et = mk(0, 0, 0);
    if (has_ks && wi_h > 0.0f) {
        V3 F = fresnel(mt.ks, wi_h);
        float G = geom(wi, N, H, rough) * geom(wo, N, H, rough);
        Es = scale((D_h * G) / (4.0f * absr(wi_n) * absr(wo_n)), F);
    }
    if (has_kt) {
        V3 At = mk(1, 1, 1);
        if (wo_n < 0) {                                                           /* sic: logf(0) = -inf is relied on */
            At.x = ort_powf(kEuler, dist * ort_logf(mt.kt.x));
            At.y = ort_powf(kEuler, dist * ort_logf(mt.kt.y));
            At.z = ort_powf(kEuler, dist * ort_logf(mt.kt.z));
        }
        if (r < 0.0f) {
            if (has_ks) Et = had(At, Es);
        } else {
            V3 F = sub(mk(1, 1, 1), fresnel(mt.ks, wi_m));
            float G = geom(wi, N, m, rough) * geom(wo, N, m, rough);
            float denom = (absr(wi_n) * absr(wo_n) * sq(bn.ni * wi_m + bn.no * wo_m));
            if (!ceq(denom, 0.0f)) {
                V3 nom = scale(D_m * G * absr(wi_m) * absr(wo_m) * sq(bn.no), F);
                Et = had(At, divs(nom, denom));
            }
        }
    }
    f = scale(absr(wi_n), add(add(mt.ed, Es), Et));
    return p;
}

/* sample_lobe (ray.cpp:1065-1091) with cos(phi), sin(phi) supplied by the caller, so that the
   kernel can evaluate the (double-precision) sine/cosine once for lanes in different states */
/* Nn = normalize(N) (ray.cpp:1069) is evaluated by the caller */
ORT_D V3 sample_lobe_n(V3 N, float c, float cos_phi, float sin_phi) {
    float s = __builtin_sqrtf(1.0f - c * c);
    V3 K = mk(s * cos_phi, s * sin_phi, c);
    if (absr(N.z - 1.0f) < 0.0001f) return K;
    if (absr(N.z + 1.0f) < 0.0001f) return mk(K.x, -K.y, -K.z);
    V3 B = normalize(mk(-N.y, N.x, 0));
    V3 C = cross(N, B);
    return add(add(scale(K.x, B), scale(K.y, C)), scale(K.z, N));
}
ORT_D V3 sample_lobe(V3 N, float c, float phi) { return sample_lobe_n(normalize(N), c, ort_cosf(phi), ort_sinf(phi)); }

/* sample_brdf (ray.cpp:1100-1161) in two halves around the evaluation of cos/sin(phi):
   draw: the three RNG draws, the lobe's cos(theta) and the azimuth phi = 2 pi e1 */
struct BrdfDraw { float c, phi, choice; };
/* DIFFUSE_SCENE: every material has pd_c = 1, so the other lobes are drawn only when the choice is exactly 1.0 -- a hint
   for the register allocator (block frequencies), not a change of the code's meaning */
template <bool DIFFUSE_SCENE = false>
ORT_D BrdfDraw sample_brdf_draw(uint32_t &rng, float rough, const Mat &mt) {
    float e0 = rng_01(rng), e1 = rng_01(rng);
    BrdfDraw d;
    d.choice = rng_01(rng);
    d.phi = 2.0f * kPi * e1;
    const bool diffuse_lobe = d.choice < mt.pd_c;
    if (DIFFUSE_SCENE ? __builtin_expect(diffuse_lobe, 1) : diffuse_lobe) d.c = __builtin_sqrtf(e0);      /* ray.cpp:1123 */
    else d.c = ort_cosf(ort_atan2f(rough * __builtin_sqrtf(e0), __builtin_sqrtf(1.0f - e0)));          /* ray.cpp:1128,1138 */
    return d;
}
/* finish: the direction from the lobe sample */
/* NORMALIZED = false leaves ray.cpp:1158's final normalisation to the caller (the kernel shares it with the
   camera branch's) */
template <bool NORMALIZED = true, bool DIFFUSE_SCENE = false>
ORT_D V3 sample_brdf_finish(V3 N, V3 Nn, V3 wo, const Mat &mt, BrdfDraw d, float cos_phi, float sin_phi, bool &is_trans) {
    float pd_c = mt.pd_c, ps_c = mt.ps_c; /* per material, ray.cpp:1105-1113 */
    V3 wi;
    is_trans = false;
    V3 m = sample_lobe_n(Nn, d.c, cos_phi, sin_phi); /* Nn = normalize(N): ray.cpp:1069 re-normalises N */
    const bool diffuse_lobe = d.choice < pd_c;
    if (DIFFUSE_SCENE ? __builtin_expect(diffuse_lobe, 1) : diffuse_lobe) {
        wi = m;
    } else {
        bool refract = !(d.choice >= pd_c && d.choice < pd_c + ps_c);
        Beer bn;
        float r = 0.0f;
        if (refract) {
            bn = beer(N, wo, mt.ior);
            r = radicand(m, wo, bn.n);
            refract = !(r < 0.0f); /* total internal reflection falls back to the mirror direction */
        }
        if (!refract) {
            wi = sub(scale(2.0f * absr(dot(wo, m)), m), wo);
        } else {
            wi = sub(scale(bn.n * dot(wo, m) - sgn(dot(wo, N)) * __builtin_sqrtf(r), m), scale(bn.n, wo));
            is_trans = true;
        }
    }
    return NORMALIZED ? normalize(wi) : wi; /* ray.cpp:1158 */
}
ORT_D V3 sample_brdf(uint32_t &rng, V3 N, V3 wo, float rough, const Mat &mt, bool &is_trans) {
    BrdfDraw d = sample_brdf_draw(rng, rough, mt);
    return sample_brdf_finish(N, normalize(N), wo, mt, d, ort_cosf(d.phi), ort_sinf(d.phi), is_trans);
}

/* ---- intersectors ----------------------------------------------------------------------- */
/* ray.cpp:63-115 with e1, e2, n = cross(e1,e2) precomputed (same f32 expressions).  Returns
   the reference's hit_t (-1 = no hit).  Evaluation order differs (u, then v, then t) but
   every value is the reference's own expression. */
ORT_D float hit_triangle(V3 v0, V3 e1, V3 e2, V3 o, V3 d) {
    V3 pv = cross(d, e2);
    float det = dot(pv, e1);
    if (!(det <= -0.000001f || det >= 0.000001f)) return -1.0f;
    V3 T = sub(o, v0);
    float u = dot(pv, T) / det;
    if (!(u >= 0.0f)) return -1.0f;
    V3 a = cross(T, e1);
    float v = dot(a, d) / det;
    if (!(v >= 0.0f && u + v <= 1.0f)) return -1.0f;
    float t = dot(a, e2) / det;
    if (!(t >= kHitTMin)) return -1.0f;
    return t;
}

/* tangent = the hit came from the "one intersection point" branch, whose t = -b/(2a) is half the
   distance to the closest approach (sic): a point outside the sphere */
ORT_D float hit_sphere(V3 c, float rad, V3 o, V3 d, V3 &n, bool &tangent) {       /* ray.cpp:132-190 */
    float ht = -1.0f;
    tangent = false;
    V3 rel = sub(o, c);
    float a = dot(d, d), b = dot(d, rel), cc = dot(rel, rel) - rad * rad;
    float root = b * b - a * cc;
    const float tol = 0.00001f;
    if (root >= tol) {
        float sr = __builtin_sqrtf(root);
        float tn = (-b - sr) / a, tp = (-b + sr) / a;
        float t = (tn < 0.0f) ? tp : tn;
        if (t > kHitTMin) {
            ht = t;
            n = scale(1.0f, sub(add(o, scale(ht, d)), c));
        }
    } else if (__builtin_expect(root < tol && root > -tol, 0)) { /* tangent: rare */
        float t = (-b) / (2 * a);
        if (t > kHitTMin) {
            ht = t;
            n = sub(add(o, scale(ht, d)), c);
            tangent = true;
        }
    }
    return ht;
}

/* ray.cpp:206-283; inv = (1/d.x, 1/d.y, 1/d.z) is the reference's own per-call value (ray.cpp:210),
   computed once per ray by the caller */
ORT_D float hit_aab(V3 lo, V3 hi, V3 o, V3 inv, V3 &n) {
    float ht = -1.0f;
    V3 t0 = had(sub(lo, o), inv), t1 = had(sub(hi, o), inv);
    V3 tmin = mk(rmin(t0.x, t1.x), rmin(t0.y, t1.y), rmin(t0.z, t1.z));
    V3 tmax = mk(rmax(t0.x, t1.x), rmax(t0.y, t1.y), rmax(t0.z, t1.z));
    float max_of_min = rmax(rmax(tmin.x, tmin.y), tmin.z);
    float min_of_max = rmin(rmin(tmax.x, tmax.y), tmax.z);
    if (min_of_max >= max_of_min) {
        float tx = t0.x, ty = t0.y, tz = t0.z;
        float sx = -1.0f, sy = -1.0f, sz = -1.0f;
        if (tx > t1.x) { tx = t1.x; sx = 1.0f; }
        if (ty > t1.y) { ty = t1.y; sy = 1.0f; }
        if (tz > t1.z) { tz = t1.z; sz = 1.0f; }
        float best = tx;
        V3 bn = mk(sx, 0, 0);
        if (best < ty) { best = ty; bn = mk(0, sy, 0); }
        if (best < tz) { best = tz; bn = mk(0, 0, sz); }
        ht = max_of_min; /* sic: may be negative; callers threshold */
        n = bn;
    }
    return ht;
}

/* hit_aab for a ray whose origin and 1/d are all finite: no product (lo - o) * (1/d) can then be a NaN, and
   without NaNs the reference's "?:" minimum / maximum (types.h:50-51) pick the same values as the hardware's
   min / max instructions (up to the sign of a zero, which no comparison below and no caller's threshold can
   tell apart) -- one instruction each instead of compare + select, and three-operand forms for the two
   reductions.  Same hit distance, same normal. */
ORT_D float hit_aab_finite(V3 lo, V3 hi, V3 o, V3 inv, V3 &n) {
    float ht = -1.0f;
    V3 t0 = had(sub(lo, o), inv), t1 = had(sub(hi, o), inv);
    V3 tmin = mk(__builtin_fminf(t0.x, t1.x), __builtin_fminf(t0.y, t1.y), __builtin_fminf(t0.z, t1.z));
    V3 tmax = mk(__builtin_fmaxf(t0.x, t1.x), __builtin_fmaxf(t0.y, t1.y), __builtin_fmaxf(t0.z, t1.z));
    float max_of_min = __builtin_fmaxf(__builtin_fmaxf(tmin.x, tmin.y), tmin.z);
    float min_of_max = __builtin_fminf(__builtin_fminf(tmax.x, tmax.y), tmax.z);
    if (min_of_max >= max_of_min) {
        float sx = (t0.x > t1.x) ? 1.0f : -1.0f, sy = (t0.y > t1.y) ? 1.0f : -1.0f, sz = (t0.z > t1.z) ? 1.0f : -1.0f;
        float best = tmin.x;
        V3 bn = mk(sx, 0, 0);
        if (best < tmin.y) { best = tmin.y; bn = mk(0, sy, 0); }
        if (best < tmin.z) { best = tmin.z; bn = mk(0, 0, sz); }
        ht = max_of_min; /* sic: may be negative; callers threshold */
        n = bn;
    }
    return ht;
}
/* hit_aab_finite's distance alone (the analytic prologue: the normal is worked out afterwards, for the winner only) */
ORT_D float hit_aab_t_finite(V3 lo, V3 hi, V3 o, V3 inv) {
    V3 t0 = had(sub(lo, o), inv), t1 = had(sub(hi, o), inv);
    float max_of_min = __builtin_fmaxf(__builtin_fmaxf(__builtin_fminf(t0.x, t1.x), __builtin_fminf(t0.y, t1.y)), __builtin_fminf(t0.z, t1.z));
    float min_of_max = __builtin_fminf(__builtin_fminf(__builtin_fmaxf(t0.x, t1.x), __builtin_fmaxf(t0.y, t1.y)), __builtin_fmaxf(t0.z, t1.z));
    return (min_of_max >= max_of_min) ? max_of_min : -1.0f;
}
/* true when every component is finite (x - x is 0 for a finite x, NaN otherwise) */
ORT_D bool all_finite6(V3 a, V3 b) {
    return (((a.x - a.x) + (a.y - a.y) + (a.z - a.z)) + ((b.x - b.x) + (b.y - b.y) + (b.z - b.z))) == 0.0f;
}

/* the same, distance only (node admission tests of the reference octree) */
ORT_D float hit_aab_t(V3 lo, V3 hi, V3 o, V3 inv) {
    V3 t0 = had(sub(lo, o), inv), t1 = had(sub(hi, o), inv);
    float max_of_min = rmax(rmax(rmin(t0.x, t1.x), rmin(t0.y, t1.y)), rmin(t0.z, t1.z));
    float min_of_max = rmin(rmin(rmax(t0.x, t1.x), rmax(t0.y, t1.y)), rmax(t0.z, t1.z));
    return (min_of_max >= max_of_min) ? max_of_min : -1.0f;
}

/* ray.cpp:286-352 with rotation_matrix_along_z(axis) (ray.cpp:8-33) and |axis| precomputed */
ORT_D float hit_cylinder(V3 base, float radius, V3 r0, V3 r1, V3 r2, float axis_len, V3 o, V3 d, V3 &n) {
    float ht = -1.0f;
    V3 rel = sub(o, base);
    o = mk(dot(r0, rel), dot(r1, rel), dot(r2, rel));
    d = mk(dot(r0, d), dot(r1, d), dot(r2, d));
    float t_bot = (-o.z) / d.z;
    float t_top = (axis_len - o.z) / d.z;
    float smin = rmin(t_bot, t_top), smax = rmax(t_bot, t_top);
    float a = d.x * d.x + d.y * d.y;
    float b = d.x * o.x + d.y * o.y;
    float c = (o.x * o.x + o.y * o.y) - radius * radius;
    float det = b * b - a * c;
    if (det >= 0.0f) {
        float sr = __builtin_sqrtf(det);
        float cmin = (-b - sr) / a, cmax = (-b + sr) / a;
        float tin = rmax(smin, cmin), tout = rmin(smax, cmax);
        if (tin <= tout) {
            ht = tin; /* sic: may be negative */
            V3 ln = mk(0, 1, 0); /* sic: cap "normal" */
            if (smin < cmin) {
                V3 p = add(o, scale(ht, d));
                ln = mk(p.x, p.y, 0);
            }
            /* transpose(rotation) * ln */
            n = mk(dot(mk(r0.x, r1.x, r2.x), ln), dot(mk(r0.y, r1.y, r2.y), ln), dot(mk(r0.z, r1.z, r2.z), ln));
        }
    }
    return ht;
}

} // namespace ortd

#endif
