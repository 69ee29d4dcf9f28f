/*
 * ort_lane.h -- the lane code of the path tracer and its kernels (gfx950), without a host side.
 *
 * One GPU lane executes one JOB at a time; a job is one call of the reference function
 * tiled_raytrace_bvh (code/ray.cpp:1178-1466): a pixel rect rendered serially with one
 * xorshift stream.  The lane is a small state machine (new job -> pixel -> sample ->
 * bounce) that alternates "produce the next ray" with an interruptible closest-hit
 * traversal; a lane whose path ends immediately starts its next sample / pixel / job.
 *
 * Traversal replaces raycast_bvh (ray.cpp:624-822): ordered depth-first walk of the
 * 2-wide tree of ort_tree.cpp with a per-lane stack whose first entries live in LDS
 * (column-per-lane, conflict-free) and whose tail spills to scratch.
 *
 * Included by two translation units: ort_kernels.hip (the library's kernels at four waves per SIMD, and the host side) and
 * ort_kernels_w5.hip (the plain-loop kernels at FIVE: 96 registers, 20 LDS stack entries, built with machine LICM off).
 * Same lane code, other limits (ORT_WAVES_PER_EU, ORT_LDS_STACK, ORT_SPILL_STACK); each unit compiles it in a namespace of its
 * own (ORT_NS), so that the two builds of one template are two symbols.  tools/host_sim.cpp compiles it for the host
 * (ORT_HOST_SIM: one simulated lane).
 */
#ifndef ORT_LANE_H
#define ORT_LANE_H

#include <hip/hip_runtime.h>

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <string>
#include <vector>

#include "ort_device.h"
#include "ort_scene.h"

#ifdef ORT_W5_TU
#define ORT_NS ort_w5
#else
#define ORT_NS ort
#endif

namespace ORT_NS {

using namespace ort;
using namespace ortd;

constexpr int kBlock = 256;      /* 4 waves */
#ifndef ORT_LDS_STACK
#define ORT_LDS_STACK 24
#endif
constexpr int kLdsStack = ORT_LDS_STACK; /* entries per lane in LDS: 24 * 256 * 4 B = 24 KB per block */
#ifndef ORT_SPILL_STACK
#define ORT_SPILL_STACK 40
#endif
constexpr int kSpillStack = ORT_SPILL_STACK;  /* scratch tail */
static_assert(kLdsStack - 4 + kSpillStack >= (int)kTreeDepthBudget, "the re-traversal of resolve_hit must hold a tree of kTreeDepthBudget levels");
constexpr uint32_t kBfsPoolQueues = 1024;     /* queues of the breadth-first fallback, shared by all lanes */
constexpr size_t kBfsPoolBytes = 1024u << 20; /* at most; a queue holds one entry per reference-tree node */
constexpr int kDiagFallback = 106;            /* fallback_counters = ctrl + 6: the diagnostics sit at ctrl[112..114] */
constexpr uint32_t kBfsLockStride = 32;       /* u32 units: every lock word has a 128-byte line to itself */

/* Small read-only tables every ray touches live in LDS, copied there once per workgroup: ~100 cycles of latency
   instead of a trip to L1 / L2 on the critical path of every ray (the kernel is latency-bound: DESIGN.md).
   Layout in float4 units; a table that does not fit its slot stays in HBM (SceneView::tab_flags). */
constexpr int kTabRoot = 0;                      /* node 0 of the fast tree (4) */
constexpr int kTabPro = 4;                       /* the analytic prologue's shapes: boxes (2 each), spheres (1), cylinders (4) */
constexpr int kTabProCap = 40;
constexpr int kTabLights = kTabPro + kTabProCap; /* light_is_sphere[64] as u32 */
constexpr int kTabLightCap = 64;
constexpr int kTabMats = kTabLights + kTabLightCap / 4; /* DevMaterial records, 5 each */
constexpr int kTabMatCap = 48;
constexpr int kTabTreelet = kTabMats + 5 * kTabMatCap; /* nodes [0, kTreeletNodes) of the fast tree, breadth-first top (ort_tree.cpp) */
constexpr int kTabF4 = kTabTreelet + 4 * (int)kTreeletNodes; /* 428 float4 = 6848 B */
enum : uint32_t { TAB_PRO = 1u, TAB_LIGHTS = 2u, TAB_MATS = 8u };

#ifdef ORT_HOST_SIM
#define ORT_CONSTANT_AS
#else
/* the structs behind these pointers are written by the host before the launch and never by a kernel: the constant
   address space tells the compiler so, and wave-uniform reads of them become scalar loads (s_load_dwordx8 ...)
   instead of per-lane vector loads of one address */
#define ORT_CONSTANT_AS __attribute__((address_space(4)))
#endif
/* The part of the scene view that only rare paths read -- the reference test order (bit-equal hit distances) and
   the exact breadth-first fallback -- lives behind one pointer in HBM: as by-value kernel arguments these twenty
   scalar registers were spilled to vector lanes and back all through the shading code. */
struct SceneCold {
    const float4 *ref_nodes;   /* reference-compatible octree: 3 per node */
    const uint32_t *ref_recs;
    const uint32_t *tri_order, *sphere_order, *box_order, *cyl_order; /* reference test order (ties) */
    /* exact fallback: a pool of queues in HBM, each long enough for every node of the reference tree
       (a ray enqueues a node at most once), taken with a try-lock for the duration of one re-cast */
    uint32_t *bfs_pool;
    uint32_t *bfs_locks;
    uint32_t bfs_queue_cap, bfs_queue_count;
    unsigned long long *fallback_counters; /* [0] rays re-cast exactly, [1] queue overflows (cannot happen: kept as a tripwire);
                                              diagnostics at [kDiagFallback]: octree nodes enqueued, rays traversed again, busy queues met */
};

/* chain: len << 28 | kChainNested | first pair of chain_boxes (ort_scene.h, RefTree); mat: material index */
struct PrimInfo { uint32_t chain, mat; };

struct SceneView {
    const float4 *tab_src;    /* kTabF4 float4, the image of the LDS tables */
    uint32_t tab_flags;
    const float4 *nodes;      /* 4 per node (DevNode), or -- WIDE kernel variants -- 8 per node (DevNode4) */
    const float4 *tris;       /* 3 per triangle: v0 e1 e2 n (12 floats) */
    const float4 *spheres;    /* 1 per sphere: c.xyz r */
    const float4 *boxes;      /* 2 per box */
    const float4 *cyls;       /* 4 per cylinder */
    /* what shading needs to know about a ray's winner: its visibility-chain word and its material index, ONE array
       over all kinds (triangles first; info_index()) so that a lane can ask for both the moment its ray is finished */
    const PrimInfo *prim_info;
    uint32_t info_box, info_cyl, info_sphere; /* first entry of each analytic kind */
    const float4 *materials;  /* 5 per material (DevMaterial) */
    const uint32_t *light_is_sphere;
    uint32_t light_count;
    uint32_t pro_boxes, pro_spheres, pro_cyls; /* analytic prologue: every ray tests shapes [0, n) of each kind outright */
    float cam[12];            /* p, x_axis, y_axis, z_axis */
    /* reference-compatible octree (ort_reftree.cpp): visibility chains */
    const float4 *chain_boxes; /* 2 per chain entry */
    const ORT_CONSTANT_AS SceneCold *cold; /* what only the rare paths read (ties, the exact fallback) */
    unsigned long long *util; /* diagnostics (ORT_DEBUG_UTIL=1, counters build): per-phase wave-iteration and active-lane sums */
    uint32_t force_fallback_mask; /* tests (ORT_DEBUG_FORCE_FALLBACK): also re-cast rays with (bits(dir.x) & mask) == 0; ~0u = off */
};

/* host: the PrimInfo array of a committed scene, triangles | boxes | cylinders | spheres */
inline void build_prim_info(const Tree &t, const RefTree &rt, std::vector<PrimInfo> &out, uint32_t &info_box, uint32_t &info_cyl, uint32_t &info_sphere) {
    info_box = (uint32_t)t.tri_mat.size();
    info_cyl = info_box + (uint32_t)t.box_mat.size();
    info_sphere = info_cyl + (uint32_t)t.cyl_mat.size();
    out.assign((size_t)info_sphere + t.sphere_mat.size(), PrimInfo{0u, 0u});
    auto fill = [&out](uint32_t base, const std::vector<uint32_t> &chain, const std::vector<uint32_t> &mat) {
        for (size_t i = 0; i < mat.size(); ++i) out[base + i] = PrimInfo{i < chain.size() ? chain[i] : 0u, mat[i]};
    };
    fill(0u, rt.tri_chain, t.tri_mat);
    fill(info_box, rt.box_chain, t.box_mat);
    fill(info_cyl, rt.cyl_chain, t.cyl_mat);
    fill(info_sphere, rt.sphere_chain, t.sphere_mat);
}

enum : int { JOBS_EXPLICIT = 0, JOBS_PIXEL = 1, JOBS_CHUNK = 2 };

struct RenderView {
    int mode;
    const ort_tile_job *jobs;
    uint32_t *final_states;
    unsigned long long job_count; /* size of the job index space */
    int W, H, x0, y0, x1, y1;
    uint32_t seed, spp, chunk, nchunks;
    float rr;
    int refill_below; /* leave the traversal loop when fewer lanes than this are still tracing */
    int descend_below; /* leave the descend loop (to process the leaves already reached, and perhaps refill) when
                          fewer lanes than this are still walking interior nodes */
    uint32_t shard_index, shard_count, blocks_w, my_blocks;
    uint32_t block_x0, block_y0; /* unsharded renders enumerate only the 8x8 blocks that touch the rect */
    float *out;      /* W*H*3, or (packed_out) this shard's blocks: my_blocks * 64 * 3 */
    float *partial;  /* CHUNK: nchunks planes of this shard's blocks, my_blocks * 64 * 3 floats each */
    int packed_out;  /* ORT_RENDER_PACKED: out holds only this shard's 8x8 blocks, [local block][pixel in block][rgb] */
    unsigned long long *next_job;
    unsigned long long *counters; /* paths rays node_tests tri_tests analytic_tests fallback_rays */
    /* ray exchange (pt_lane_x): every wave owns two LIFO stashes in HBM, L for parked paths whose ray is still
       being traversed and R for parked paths whose ray is finished; float4 units */
    float4 *stash;
    uint32_t stash_wave_f4; /* per wave: L records kStashVecs * capL, L stacks kLdsStack / 4 * capL, R records kStashVecs * capR (device_render sizes it) */
    uint32_t capL, capR;
    uint32_t long_min;   /* start a traversal phase on parked rays when tracing lanes + parked rays reach this */
    uint32_t long_refill; /* within such a phase, take more parked rays when fewer lanes than this are tracing */
    uint32_t park_min;   /* stragglers are parked only when there are at least this many of them (fewer: they idle through one shading pass, cheaper than an exchange step) */
    uint32_t inflight_cap; /* a lane without a path starts a new job only while the wave holds fewer parked paths than this
                              (every parked path is a job in progress: the more a wave holds, the longer its tail) */
    uint32_t block_major; /* CHUNK policy: the job space is [block][chunk][pixel] (see below) instead of [chunk][block][pixel] */
    unsigned long long endgame_from; /* ray exchange: job index from which waves stop parking and drain their stashes (pt_lane_x) */
    uint32_t job_batch;  /* a wave draws this many job indices from next_job at a time and hands them to its lanes one by one (0 / 1: every draw goes to next_job) */
    unsigned long long batch_until; /* ... while the job counter, as the wave last saw it, is below this index; after it: exactly as many as it needs */
    unsigned long long *drain; /* diagnostics (ORT_DEBUG_DRAIN): when each wave ran out of work (s_memrealtime), [workgroup * 4 + wave] */
};

/* ---- the order in which a CHUNK render issues its jobs ----------------------------------------------------------
 * A job is a serial stream of `chunk` samples that only one lane can advance.  The policy defines the job SET -- chunk k
 * of pixel i, seeded by (k, i) -- not an order, and seeds belong to jobs, so the order of issue cannot change a bit of
 * the image.  Issued chunk-major ([chunk][block][pixel], rounds 1-2) a launch ended with a sweep over the whole image in
 * which the last lanes to draw an expensive job (a bunny pixel costs eight wall pixels) finished it alone: 12-19 ms on
 * the 57 ms an 8-way shard of the headline frame needs (profiles/r03_scaling_proxy.json, r03_tuning.md).  Now
 * BLOCK-major, [block][chunk][pixel]: all chunks of an 8x8 block are issued together, so the lanes of a wave work on the
 * same few pixels with different seeds -- like rays, like costs, jobs that end together -- and the launch ends on its last
 * BLOCKS, not on a last pass over everything: 8-way shard 70.8 -> 64.6 ms, plain loop on the whole frame 496 -> 461 ms.
 * Issuing the blocks most expensive first on top of that (longest processing time first) was built twice and bought
 * nothing: measured and sorted inside the launch, the lists are never ready in time (the expensive blocks are exactly the
 * ones whose measuring jobs end last); measured by one render and used by the next, an 8-way shard took 65.6 ms against
 * 65.2 in natural block order (profiles/r03_tuning.md).  Natural block order it is: no state, no atomics. */

/* What the kernels receive by value: the handful of render parameters every ray reads; everything else stays in the
   RenderView in HBM behind `c` (job decoding, pixel addresses, stashes: read once per job or per pixel).  By value the
   whole RenderView cost ~45 scalar registers, spilled to vector lanes and back all through the lane code. */
struct RenderHot {
    int mode, W, H;
    float rr;
    int refill_below, descend_below;
    const ORT_CONSTANT_AS RenderView *c;
};

/* wavefront mode: per-slot path state in HBM, structure-of-arrays so that a wave's loads and
   stores are coalesced (consecutive lanes = consecutive slots); 112 B per slot */
struct WfView {
    uint32_t slots;
    float4 *od0;     /* org.xyz dir.x            (shade -> trace) */
    float2 *od1;     /* dir.y dir.z */
    float4 *hit0;    /* best_t hit_n.xyz         (trace -> shade) */
    uint32_t *hitp;  /* hit_prim */
    float4 *p0;      /* weight.xyz color.x       (shade -> shade) */
    float4 *p1;      /* color.yz wo.xy */
    float4 *p2;      /* wo.z rng sample spp */
    uint4 *p3;       /* job_index  px|py<<16  jx0|jx1<<16  jy1|plane<<16 */
    uint32_t *flags; /* bits 0-2 ps, bit 3 primary, bit 4 has a ray to trace */
    unsigned long long *active; /* rays produced by the last counted shade launch */
};

/* ---- kernel ---------------------------------------------------------------------------- */
enum : int { PS_NEED_JOB = 0, PS_PIXEL = 1, PS_SAMPLE = 2, PS_HIT = 3, PS_DONE = 4 };

#ifdef ORT_HOST_SIM /* one simulated lane per host thread: tools/host_sim.cpp (job draws and counters are atomic: its threads share them) */
#define ORT_BALLOT(p) ((p) ? 1ull : 0ull)
#define ORT_POPC64(m) __builtin_popcountll(m)
#define ORT_NEXT_JOB(p) __atomic_fetch_add((p), 1ull, __ATOMIC_RELAXED)
#define ORT_COUNT(p, v) ((void)__atomic_fetch_add((p), (v), __ATOMIC_RELAXED))
#define ORT_TRY_LOCK(p) (*(p) == 0u ? (*(p) = 1u, true) : false)
#define ORT_PEEK(p) (*(p))
#define ORT_BACKOFF()
#define ORT_UNLOCK(p) (*(p) = 0u)
#define ORT_FENCE()
#define ORT_FFS64(m) __builtin_ffsll((long long)(m))
#define ORT_LANE() 0
#define ORT_UTIL(sv, k, pred)
#define ORT_PHASE(pr, sv, k, pred)
#ifndef ORT_SIM_PIXEL_HOOK
#define ORT_SIM_PIXEL_HOOK(x, y, rng)
#endif
#ifndef ORT_SIM_RAY_HOOK
#define ORT_SIM_RAY_HOOK(x, y, o, d, t, n, m)
#endif
#else
#define ORT_SIM_PIXEL_HOOK(x, y, rng)
#define ORT_SIM_RAY_HOOK(x, y, o, d, t, n, m)
#define ORT_BALLOT(p) __ballot(p)
#define ORT_POPC64(m) __popcll(m)
#define ORT_NEXT_JOB(p) atomicAdd((p), 1ull)
#define ORT_COUNT(p, v) atomicAdd((p), (v))
#define ORT_TRY_LOCK(p) (atomicCAS((p), 0u, 1u) == 0u)
#define ORT_PEEK(p) __hip_atomic_load((p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#define ORT_BACKOFF() __builtin_amdgcn_s_sleep(64) /* 64 x 64 clocks, ~2 us */
#define ORT_UNLOCK(p) ((void)atomicExch((p), 0u))
#define ORT_FENCE() __threadfence()
#define ORT_FFS64(m) __ffsll((unsigned long long)(m))
#define ORT_LANE() ((int)__lane_id())
/* lane-utilisation probe: event k happened in this wave with popc(pred) lanes taking part.  Diagnostics build only
   (COUNTERS, ORT_DEBUG_UTIL=1); the waves of the first 32 workgroups record, into LDS (global atomics here would
   keep every following load waiting behind them), flushed to memory when the workgroup ends */
#define ORT_UTIL(sv, k, pred)                                                                        \
    do {                                                                                             \
        if (COUNTERS && (sv).util && blockIdx.x < 32u) {                                             \
            unsigned long long m_ = __ballot(pred);                                                  \
            if (m_ && (int)__lane_id() == __ffsll(m_) - 1) {                                         \
                atomicAdd(&g_lds_prof[2 * (k)], 1ull);                                               \
                atomicAdd(&g_lds_prof[2 * (k) + 1], (unsigned long long)__popcll(m_));               \
            }                                                                                        \
        }                                                                                            \
    } while (0)
/* phase timer (same diagnostics build): the shader cycles since the wave's previous mark are charged to phase k,
   with the number of lanes for which pred holds */
#define ORT_PHASE(pr, sv, k, pred)                                                                   \
    do {                                                                                             \
        if (COUNTERS && (pr).on) {                                                                   \
            const unsigned long long now_ = __builtin_amdgcn_s_memtime();                            \
            const unsigned long long m_ = __ballot(pred), a_ = __ballot(true);                       \
            /* the wave's previous mark lives in LDS: a register would only be updated in the lanes active there */ \
            unsigned long long *last_ = &g_lds_prof[96 + (threadIdx.x >> 6)];                        \
            if ((int)__lane_id() == __ffsll(a_) - 1) {                                               \
                atomicAdd(&g_lds_prof[32 + 3 * (k)], now_ - *last_);                                 \
                atomicAdd(&g_lds_prof[32 + 3 * (k) + 1], 1ull);                                      \
                atomicAdd(&g_lds_prof[32 + 3 * (k) + 2], (unsigned long long)__popcll(m_));          \
                *last_ = __builtin_amdgcn_s_memtime();                                               \
            }                                                                                        \
        }                                                                                            \
    } while (0)
#endif

#ifndef ORT_HOST_SIM
__shared__ unsigned long long g_lds_prof[96 + 4]; /* + the four waves' previous marks */ /* diagnostics build only: [0,32) event probes, [32,96) phase timers */
#endif

struct Prof { unsigned long long t = 0; bool on = false; };

#ifndef ORT_DESCEND_SHIFT
#define ORT_DESCEND_SHIFT 2
#endif

/* Branch-frequency hints on the exactness machinery of resolve_hit (chain walk of odd chains, phantom hits, re-traversals,
   the exact fallback): the register allocator weighs spill code by block frequency and this kernel lives at its
   128-register cap, so telling it that these paths are rare moves the spills there: +4.4 % on the bunny room.  (Hints on
   ties, deep stacks and the once-per-job blocks were neutral or harmful and are not kept.) */
#define ORT_RARE(x) __builtin_expect(!!(x), 0)
#if defined(ORT_HOST_SIM) && defined(ORT_CHAIN_STATS) /* tools/host_sim: why rays leave the fast path */
extern unsigned long long g_cs[4][16];
#define ORT_STAT(row, col) (g_cs[row][col]++)
#else
#define ORT_STAT(row, col) ((void)0)
#endif
constexpr uint32_t kNoPrim = 0xffffffffu;
constexpr uint32_t kTraversalDone = 0xffffffffu; /* == EMPTY_CHILD: a leaf word no tree contains */

ORT_D uint32_t prim_order(const SceneView &sv, uint32_t kind, uint32_t slot) {
    return (kind == PRIM_TRI) ? sv.cold->tri_order[slot] : (kind == PRIM_SPHERE) ? sv.cold->sphere_order[slot]
         : (kind == PRIM_BOX) ? sv.cold->box_order[slot] : sv.cold->cyl_order[slot];
}

/* one primitive against the ray, exactly as raycast_bvh does per record (ray.cpp:647-716):
   accept when hit_t >= 1e-6 and strictly closer than the best so far */
template <bool COUNTERS, bool EXACT_ORDER, bool FINITE_RAY = false, bool FROM_TAB = false>
ORT_D void test_prim(const SceneView &sv, uint32_t kind, uint32_t slot, V3 org, V3 dir, V3 inv_d, float &best_t, V3 &hit_n,
                     uint32_t &hit_prim, float &phantom_t, float &runner_t, unsigned long long &c_tris, unsigned long long &c_analytic,
                     uint32_t excl = 0xffffffffu, const float4 *rec = nullptr /* FROM_TAB: the shape's record in the LDS tables */) {
    float t;
    V3 n = mk(0, 0, 0);
    bool tangent = false;
    if (!EXACT_ORDER && ((kind << 28) | slot) == excl) return;
    if (kind == PRIM_TRI) {
        const float4 *tp = sv.tris + 3u * slot;
        float4 a = tp[0], b = tp[1], c = tp[2];
        if (COUNTERS) c_tris++;
        t = hit_triangle(mk(a.x, a.y, a.z), mk(a.w, b.x, b.y), mk(b.z, b.w, c.x), org, dir);
        n = mk(c.y, c.z, c.w);
    } else if (kind == PRIM_SPHERE) {
        float4 s;
        if (FROM_TAB) s = rec[0]; else s = sv.spheres[slot];
        if (COUNTERS) c_analytic++;
        t = hit_sphere(mk(s.x, s.y, s.z), s.w, org, dir, n, tangent);
    } else if (kind == PRIM_BOX) {
        float4 lo, hi;
        if (FROM_TAB) { lo = rec[0]; hi = rec[1]; } else { lo = sv.boxes[2u * slot]; hi = sv.boxes[2u * slot + 1u]; }
        if (COUNTERS) c_analytic++;
        t = FINITE_RAY ? hit_aab_finite(mk(lo.x, lo.y, lo.z), mk(hi.x, hi.y, hi.z), org, inv_d, n)
                       : hit_aab(mk(lo.x, lo.y, lo.z), mk(hi.x, hi.y, hi.z), org, inv_d, n);
    } else {
        float4 a, b, c, d;
        if (FROM_TAB) { a = rec[0]; b = rec[1]; c = rec[2]; d = rec[3]; }
        else { const float4 *cp = sv.cyls + 4u * slot; a = cp[0]; b = cp[1]; c = cp[2]; d = cp[3]; }
        if (COUNTERS) c_analytic++;
        t = hit_cylinder(mk(a.x, a.y, a.z), a.w, mk(b.x, b.y, b.z), mk(b.w, c.x, c.y), mk(c.z, c.w, d.x), d.y, org, dir, n);
    }
    if (!EXACT_ORDER && tangent) {
        /* a phantom hit outside its box: whether the reference sees it depends on its visiting
           order, so it never competes here; the caller re-casts the ray exactly if it could win */
        phantom_t = fminf(phantom_t, t);
        return;
    }
    bool take = (t >= kHitTMin && t < best_t);
    if (!EXACT_ORDER && t == best_t && t >= kHitTMin && hit_prim != kNoPrim) {
        /* bit-equal distance (e.g. the shared diagonal of a fan-triangulated quad): the reference
           keeps whichever it tested first */
        take = prim_order(sv, kind, slot) < prim_order(sv, hit_prim >> 28, hit_prim & 0x00ffffffu);
    }
    /* the nearest hit that does NOT win (fast traversal only): resolve_hit needs to know that nothing
       else lies between the winner and the entry of its node boxes */
    if (!EXACT_ORDER && t >= kHitTMin) runner_t = fminf(runner_t, take ? best_t : t);
    if (take) {
        best_t = t;
        hit_n = n;
        hit_prim = (kind << 28) | slot;
    }
}

/* the reference's child test (ray.cpp:788-803) as far as it can be decided after the fact: origin
   inside the box (half-open), or the slab test enters at 1e-6 <= t <= t_hit, t_hit being the distance of
   the winner W of the fast traversal (the minimum over ALL primitives, ties to the lower test rank).
   The reference's clause is "t < best at that moment"; every primitive it tested before this node has a
   lower rank than W, hence a strictly larger distance, so best > t_hit and an entry at t <= t_hit
   passes.  An entry BEYOND t_hit (a hit in front of its own node box: cylinder and sphere boxes are
   not conservative to the last ulp, and a flat box around an axis-aligned triangle rounds either way)
   passes or not depending on what was found earlier -- unless nothing else CAN have been found below it:
   t_other is the nearest other hit (runner-up or phantom), exact within 2e-4 of t_hit because the fast
   traversal tests everything in that window; an entry below t_other (and inside the window) is below
   any best the reference can have held.  Otherwise undecidable here: the caller re-casts exactly. */
enum : int { CH_ADMIT = 0, CH_REJECT = 1, CH_UNKNOWN = 2 };

ORT_D uint32_t info_index(const SceneView &sv, uint32_t prim) {
    const uint32_t kind = prim >> 28, slot = prim & 0x00ffffffu;
    return slot + ((kind == PRIM_TRI) ? 0u : (kind == PRIM_BOX) ? sv.info_box : (kind == PRIM_CYL) ? sv.info_cyl : sv.info_sphere);
}

/* A lane whose ray is finished knows its winner long before the wave gets to shade it (the other lanes are still
   traversing): it asks for the winner's PrimInfo right away, with two loads that have no register destination
   (LDS-DMA, global_load_lds_dword) and land in entries 0 and 1 of the lane's own, now idle, traversal stack.
   resolve_hit picks them up after an s_waitcnt vmcnt(0): two dependent round trips (chain word, material index)
   less on the critical path of every shading pass.  hipcc does not count these loads; its own waits only become
   conservative by them (memory returns in order). */
template <int BLOCK>
ORT_D void announce_winner(const SceneView &sv, uint32_t prim, uint32_t *lds_stack, int tid) {
    if (prim == kNoPrim) return;
    const PrimInfo *p = sv.prim_info + info_index(sv, prim);
#ifdef ORT_HOST_SIM
    lds_stack[tid] = p->chain;
    lds_stack[BLOCK + tid] = p->mat;
#else
    /* M0 = LDS byte address of the wave's 64 consecutive words of one stack entry; lane i lands at M0 + 4 i */
    const uint32_t dst = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(uintptr_t)(lds_stack + (tid & ~63)));
    const uint32_t *q = &p->mat;
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\t"
                 "s_add_u32 m0, %3, %4\n\ts_nop 0\n\tglobal_load_lds_dword %2, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(p), "v"(q), "s"(dst), "n"(BLOCK * 4) : "scc");
#endif
}

/* the announced words of this lane (see announce_winner) */
template <int BLOCK>
ORT_D void announced_info(const uint32_t *lds_stack, int tid, uint32_t &chain, uint32_t &mat) {
#ifndef ORT_HOST_SIM
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
    chain = lds_stack[tid];
    mat = lds_stack[BLOCK + tid];
}
/* CH_REJECT: never admitted whatever was found before (the ray misses the box or enters below 1e-6 from
   outside): the shapes below are invisible to this ray.  CH_UNKNOWN: enters at t_entry > t_hit with another
   hit possibly in between; t_entry is handed back in gap */
ORT_D int ref_node_verdict(V3 lo, V3 hi, V3 org, V3 inv_d, float t_hit, float t_other, float &gap) {
    if ((org.x >= lo.x && org.x < hi.x) && (org.y >= lo.y && org.y < hi.y) && (org.z >= lo.z && org.z < hi.z)) return CH_ADMIT;
    const float t = hit_aab_t(lo, hi, org, inv_d);
    if (!(t >= kHitTMin)) return CH_REJECT;
    if (t <= t_hit || (t < t_other && t < t_hit * 1.0001f)) return CH_ADMIT;
    gap = fmaxf(gap, t);
    return CH_UNKNOWN;
}

/* would the reference have reached this primitive?  Every node box on the way down must admit the
   ray (origin inside, half-open; or entered at t >= 1e-6).  Entries run from the primitive's own node
   (entry 0, the smallest box) up to the root's child.  chain_verdict_full tests them all, four at a
   time so the (divergent, L2-latency-bound) loads overlap. */
ORT_D int chain_verdict_full(const SceneView &sv, uint32_t first, uint32_t len, V3 org, V3 inv_d, float t_hit, float t_other, float &gap) {
    int verdict = CH_ADMIT;
    for (uint32_t base = 0; base < len; base += 4u) {
        float4 lo[4], hi[4];
#pragma unroll
        for (uint32_t k = 0; k < 4u; ++k) {
            uint32_t i = base + k;
            i = (i < len) ? i : (len - 1u); /* clamp: re-tests the last entry, harmless */
            lo[k] = sv.chain_boxes[2u * (first + i)];
            hi[k] = sv.chain_boxes[2u * (first + i) + 1u];
        }
#pragma unroll
        for (uint32_t k = 0; k < 4u; ++k) {
            const int v = ref_node_verdict(mk(lo[k].x, lo[k].y, lo[k].z), mk(hi[k].x, hi[k].y, hi[k].z), org, inv_d, t_hit, t_other, gap);
            verdict = (v == CH_REJECT || verdict == CH_REJECT) ? CH_REJECT : (v == CH_UNKNOWN ? CH_UNKNOWN : verdict);
        }
        if (verdict == CH_REJECT) break;
    }
    return verdict;
}

ORT_D bool in_rect_half_open(float4 lo, float4 hi, V3 org) { /* math.h:1156-1169 */
    return (org.x >= lo.x && org.x < hi.x) && (org.y >= lo.y && org.y < hi.y) && (org.z >= lo.z && org.z < hi.z);
}

/* The octree's boxes are nested (a shape grows every node it is pushed through, ray.cpp:1799-1948;
   ort_reftree.cpp verifies it per chain and sets kChainNested).  With nested boxes B_top >= ... >= B_0
   and a finite 1/d the admission tests are monotone: "origin inside" can only turn false going down, the
   entry distance max_a min(t_lo, t_hi) can only grow (each per-axis near distance is a monotone float
   function of the box bound) and the exit distance can only shrink.  So with B_j the first box from the
   top that does not contain the origin, the whole chain admits the ray  <=>  B_j and B_0 do: boxes above
   B_j contain the origin; boxes between are entered no earlier than B_j and are hit if B_0 is.
   Two slab tests instead of one per level. */
constexpr uint32_t kChainNested = 0x08000000u;
#if defined(ORT_HOST_SIM) && defined(ORT_CHAIN_CROSSCHECK)
static unsigned long long g_chain_crosschecks = 0;
#endif
ORT_D int chain_verdict(const SceneView &sv, uint32_t word, V3 org, V3 inv_d, float t_hit, float t_other, float &gap) {
#ifdef ORT_MEASURE_NO_CHAIN /* developer measurement only (wrong images): what the visibility-chain check costs, by leaving it out */
    gap = 0.0f;
    return CH_ADMIT;
#endif
    const uint32_t len = word >> 28, first = word & 0x07ffffffu;
    gap = 0.0f;
    if (len == 0u) return CH_ADMIT;
    const bool finite = (om_f32_bits(inv_d.x) & 0x7fffffffu) < 0x7f800000u && (om_f32_bits(inv_d.y) & 0x7fffffffu) < 0x7f800000u &&
                        (om_f32_bits(inv_d.z) & 0x7fffffffu) < 0x7f800000u;
    if (ORT_RARE(!(word & kChainNested) || !finite)) return chain_verdict_full(sv, first, len, org, inv_d, t_hit, t_other, gap);
    const float4 dlo = sv.chain_boxes[2u * first], dhi = sv.chain_boxes[2u * first + 1u];
    float4 jlo = dlo, jhi = dhi;
    bool found = false;
#ifndef ORT_CHAIN_ROUND
#define ORT_CHAIN_ROUND 2 /* boxes fetched per step of the (rare) scan below */
#endif
    if (len >= 2u) {
        /* ONE round of loads settles nearly every chain: the leaf box (above), its parent (entry 1), the top two
           ancestors (entries len-1, len-2).  Origin inside the parent: inside every ancestor (nested), none to enter.
           Otherwise the first box from the top that does not contain it is the top one, the next, or -- a long chain
           with the origin inside its top two boxes -- found by scanning on down; the parent at the latest. */
        const uint32_t it = len - 1u, it1 = (len >= 3u) ? len - 2u : 1u;
        const float4 plo = sv.chain_boxes[2u * (first + 1u)], phi = sv.chain_boxes[2u * (first + 1u) + 1u];
        const float4 tlo = sv.chain_boxes[2u * (first + it)], thi = sv.chain_boxes[2u * (first + it) + 1u];
        const float4 ulo = sv.chain_boxes[2u * (first + it1)], uhi = sv.chain_boxes[2u * (first + it1) + 1u];
        if (!in_rect_half_open(plo, phi, org)) {
            found = true;
            if (!in_rect_half_open(tlo, thi, org)) { jlo = tlo; jhi = thi; }
            else if (!in_rect_half_open(ulo, uhi, org)) { jlo = ulo; jhi = uhi; }
            else {
                jlo = plo; jhi = phi;
                bool hit = false;
                for (int32_t top = (int32_t)len - 3; ORT_RARE(!hit && top >= 2); top -= ORT_CHAIN_ROUND) {
                    float4 lo[ORT_CHAIN_ROUND], hi[ORT_CHAIN_ROUND];
#pragma unroll
                    for (int32_t k = 0; k < ORT_CHAIN_ROUND; ++k) {
                        int32_t i = top - k;
                        i = (i > 2) ? i : 2; /* clamp: re-tests entry 2, harmless */
                        lo[k] = sv.chain_boxes[2u * (first + (uint32_t)i)];
                        hi[k] = sv.chain_boxes[2u * (first + (uint32_t)i) + 1u];
                    }
#pragma unroll
                    for (int32_t k = 0; k < ORT_CHAIN_ROUND; ++k) {
                        const bool outside = !in_rect_half_open(lo[k], hi[k], org);
                        if (!hit && outside) { jlo = lo[k]; jhi = hi[k]; hit = true; }
                    }
                }
            }
        }
    }
#if defined(ORT_HOST_SIM) && defined(ORT_CHAIN_STATS)
    { int jj = -1; for (int32_t i = (int32_t)len - 1; i >= 1; --i) if (!in_rect_half_open(sv.chain_boxes[2u*(first+i)], sv.chain_boxes[2u*(first+i)+1u], org)) { jj = (int)len - 1 - i; break; }
      g_cs[0][len]++; g_cs[1][jj < 0 ? 15 : jj]++; }
#endif
    /* found: an ancestor that does not contain the origin must be entered at t >= 1e-6.  Then, and when every
       ancestor contains the origin, the leaf box decides the rest (origin inside it, or the bounds on its entry
       distance): the entry distance only grows down the chain */
    int verdict = CH_ADMIT;
    if (found && !(hit_aab_t(mk(jlo.x, jlo.y, jlo.z), mk(jhi.x, jhi.y, jhi.z), org, inv_d) >= kHitTMin)) verdict = CH_REJECT;
    else verdict = ref_node_verdict(mk(dlo.x, dlo.y, dlo.z), mk(dhi.x, dhi.y, dhi.z), org, inv_d, t_hit, t_other, gap);
#if defined(ORT_HOST_SIM) && defined(ORT_CHAIN_CROSSCHECK) /* tools/host_sim: the shortcut against the full walk, every ray */
    {
        float g2 = 0.0f;
        if (verdict != chain_verdict_full(sv, first, len, org, inv_d, t_hit, t_other, g2)) { fprintf(stderr, "chain shortcut disagrees with the full walk\n"); abort(); }
        g_chain_crosschecks++;
    }
#endif
    return verdict;
}

/* exact fallback: raycast_bvh (ray.cpp:624-822) emulated literally on the reference-compatible
   octree -- breadth-first, children in slot order, records in push order, a child admitted when
   the origin is inside it or 1e-6 <= t_entry < best AT THAT MOMENT.  The reference never reuses
   queue memory within a ray; the emulation appends to one queue of the pool in HBM (a ray enqueues a node
   at most once, so ref_node_count entries are enough).
   Returns false if the queue overflowed (the render call then fails). */
template <bool COUNTERS>
ORT_D bool ref_raycast_bfs(const SceneView &sv, V3 org, V3 dir, V3 inv_d, uint32_t *queue, float &best_t, V3 &hit_n,
                           uint32_t &hit_prim, unsigned long long &c_nodes, unsigned long long &c_tris,
                           unsigned long long &c_analytic) {
    best_t = 3.402823466e+38f;
    hit_n = mk(0, 0, 0);
    hit_prim = kNoPrim;
    float unused = 0;
    uint32_t head = 0, tail = 0;
    bool ok = true;
    const uint32_t cap = sv.cold->bfs_queue_cap;
    queue[tail++] = 0;
    while (head != tail) {
        uint32_t node = queue[head++];
        const float4 *np = sv.cold->ref_nodes + 3u * node;
        float4 a = np[0], b = np[1], c = np[2];
        int32_t first_child = (int32_t)om_f32_bits(a.w);
        uint32_t rec_first = om_f32_bits(b.w), rec_count = om_f32_bits(c.x);
        for (uint32_t r = 0; r < rec_count; ++r) {
            uint32_t rec = sv.cold->ref_recs[rec_first + r];
            test_prim<COUNTERS, true>(sv, rec >> 28, rec & 0x00ffffffu, org, dir, inv_d, best_t, hit_n, hit_prim, unused, unused, c_tris, c_analytic);
        }
        if (first_child >= 0) {
            for (uint32_t k = 0; k < 8u; ++k) {
                uint32_t ci = (uint32_t)first_child + k;
                const float4 *cp = sv.cold->ref_nodes + 3u * ci;
                float4 ca = cp[0], cb = cp[1], cc = cp[2];
                uint32_t flags = om_f32_bits(cc.y);
                bool leaf_with_records = (flags & 3u) == 3u;
                bool has_children = (int32_t)om_f32_bits(ca.w) >= 0;
                if (!(leaf_with_records || has_children)) continue;
                V3 lo = mk(ca.x, ca.y, ca.z), hi = mk(cb.x, cb.y, cb.z);
                bool add = (org.x >= lo.x && org.x < hi.x) && (org.y >= lo.y && org.y < hi.y) && (org.z >= lo.z && org.z < hi.z);
                if (!add) {
                    float t = hit_aab_t(lo, hi, org, inv_d);
                    if (COUNTERS) c_nodes++;
                    add = (t >= kHitTMin && t < best_t);
                }
                if (add) {
                    if (tail >= cap) { ok = false; continue; }
                    queue[tail++] = ci;
                }
            }
        }
    }
    ORT_COUNT(sv.cold->fallback_counters + kDiagFallback, (unsigned long long)tail); /* diagnostics (ORT_DEBUG_FALLBACK): nodes enqueued */
    return ok;
}

/* ---- the lane: path state, hit resolution, ray production, traversal ------------------------
 * Shared by the two execution modes:
 *   persistent: one kernel, every lane loops  produce_ray <-> traverse  (pt_persistent)
 *   wavefront : path state lives in HBM; wf_shade runs produce_ray once per slot, wf_trace
 *               runs traverse + resolve_hit once per slot, alternating over all slots. */
struct Counters {
    unsigned long long paths = 0, rays = 0, nodes = 0, tris = 0, analytic = 0;
};

/* job bookkeeping is packed (image sizes and chunk counts fit 16 bits; checked on the host) so that
   few registers stay live across the traversal loop */
struct PathState {
    int ps = PS_NEED_JOB;
    uint32_t rng = 0, job_index = 0;
    uint32_t pxy = 0;  /* px | py << 16: the pixel being rendered */
    uint32_t jxx = 0;  /* jx0 | jx1 << 16: the job rect's x range */
    uint32_t jyp = 0;  /* jy1 | plane << 16: the rect's end row; CHUNK policy: which partial plane */
    uint32_t spp = 0, sample = 0;
    V3 color, org, dir, wo, weight;
    bool primary = true;
};

struct HitState {
    float best_t = 0;
    V3 hit_n;
    uint32_t hit_prim = kNoPrim;
    float phantom_t = 0;
    float runner_t = 0; /* nearest hit other than the winner, exact below best_t * 1.0002 (kCullSlack) */
    uint32_t hit_mat = 0; /* material index of hit_prim, 0 = no hit: set by resolve_hit */
};

/* position of pixel (x, y) in this shard's packed block layout [local block][pixel in block]: blocks are numbered
   row-major over the block grid and dealt round-robin, so the shard's k-th block is block shard_index + k * shard_count */
ORT_D size_t packed_index(const RenderHot &rv, uint32_t x, uint32_t y) {
    const uint32_t blk = ((y >> 3) - rv.c->block_y0) * rv.c->blocks_w + ((x >> 3) - rv.c->block_x0);
    return (size_t)((blk - rv.c->shard_index) / rv.c->shard_count) * 64u + ((y & 7u) << 3) + (x & 7u);
}
/* where a job writes pixel (x, y): its partial plane (CHUNK; packed, so a shard keeps 1/N of a frame per plane) or
   the output image (full frame, or packed on request) */
ORT_D float *pixel_ptr(const RenderHot &rv, uint32_t plane, uint32_t x, uint32_t y) {
    if (rv.mode == JOBS_CHUNK) return rv.c->partial + ((size_t)plane * rv.c->my_blocks * 64u + packed_index(rv, x, y)) * 3u;
    if (rv.c->packed_out) return rv.c->out + packed_index(rv, x, y) * 3u;
    return rv.c->out + 3u * ((size_t)y * (size_t)rv.W + (size_t)x);
}

/* traversal state of one ray on the fast tree */
struct Trav {
    uint32_t cur = 0;
    int sp = 0;
    V3 inv_d;
};

ORT_D void reset_hit(HitState &h, float best_t) {
    h.best_t = best_t;
    h.hit_n = mk(0, 0, 0);
    h.hit_prim = kNoPrim;
    h.phantom_t = __builtin_inff(); /* none yet; compared with <= against best_t (<= FLT_MAX) */
    h.runner_t = __builtin_inff();
}

/* the analytic prologue (ort_tree.cpp): the lanes that start a ray now all test the same shape at the same
   time -- uniform addresses, no divergence -- and enter the tree with best_t already set */
/* HAS_EXCL: the re-traversals of resolve_hit ignore one shape (excl); a ray's first traversal ignores none */
template <bool COUNTERS, bool TABS, bool HAS_EXCL = false>
ORT_D void prologue_tests(const SceneView &sv, const float4 *tab, V3 org, V3 dir, V3 inv_d, HitState &h, Counters &c, uint32_t excl = kNoPrim) {
    /* TABS: the shapes' records come from the LDS tables */
    const float4 *pb = tab + kTabPro, *ps = pb + 2u * sv.pro_boxes, *pc = ps + sv.pro_spheres;
    /* boxes: when every lane's origin and 1/d are finite (all but a handful of rays), the slab test runs on the
       hardware's min / max (hit_aab_finite: same values); wave-uniform choice, so no lane waits for the other form */
#ifndef ORT_PROLOGUE_DEFER
#define ORT_PROLOGUE_DEFER 1 /* 0: every box test works out its normal (A/B builds; same results) */
#endif
    if (!ORT_PROLOGUE_DEFER && ORT_BALLOT(!all_finite6(org, inv_d)) == 0ull) {
        for (uint32_t i = 0; i < sv.pro_boxes; ++i)
            test_prim<COUNTERS, false, true, TABS>(sv, PRIM_BOX, i, org, dir, inv_d, h.best_t, h.hit_n, h.hit_prim, h.phantom_t, h.runner_t, c.tris, c.analytic, excl, pb + 2u * i);
    } else
    if (ORT_BALLOT(!all_finite6(org, inv_d)) == 0ull) {
        /* distances only while the boxes compete (test_prim's rule: accept 1e-6 <= t < best, equal distances by reference
           test order, the runner-up kept); the entering face's normal is worked out once, below, for the box that won */
        for (uint32_t i = 0; i < sv.pro_boxes; ++i) {
            float4 lo, hi;
            if (TABS) { lo = pb[2u * i]; hi = pb[2u * i + 1u]; } else { lo = sv.boxes[2u * i]; hi = sv.boxes[2u * i + 1u]; }
            const uint32_t prim = ((uint32_t)PRIM_BOX << 28) | i;
            if (COUNTERS && (!HAS_EXCL || prim != excl)) c.analytic++;
            float t = hit_aab_t_finite(mk(lo.x, lo.y, lo.z), mk(hi.x, hi.y, hi.z), org, inv_d);
            if (HAS_EXCL && ORT_RARE(prim == excl)) t = -1.0f;
            bool take = (t >= kHitTMin && t < h.best_t);
            if (ORT_RARE(t == h.best_t && t >= kHitTMin && h.hit_prim != kNoPrim))
                take = prim_order(sv, PRIM_BOX, i) < prim_order(sv, h.hit_prim >> 28, h.hit_prim & 0x00ffffffu);
            if (t >= kHitTMin) h.runner_t = fminf(h.runner_t, take ? h.best_t : t);
            if (take) { h.best_t = t; h.hit_prim = prim; }
        }
        if ((h.hit_prim >> 28) == PRIM_BOX && h.hit_prim != kNoPrim) { /* reset_hit precedes every prologue: a box winner here is one of these */
            const uint32_t i = h.hit_prim & 0x00ffffffu;
            float4 lo, hi;
            if (TABS) { lo = pb[2u * i]; hi = pb[2u * i + 1u]; } else { lo = sv.boxes[2u * i]; hi = sv.boxes[2u * i + 1u]; }
            V3 n = mk(0, 0, 0);
            (void)hit_aab_finite(mk(lo.x, lo.y, lo.z), mk(hi.x, hi.y, hi.z), org, inv_d, n);
            h.hit_n = n;
        }
    } else {
        for (uint32_t i = 0; i < sv.pro_boxes; ++i)
            test_prim<COUNTERS, false, false, TABS>(sv, PRIM_BOX, i, org, dir, inv_d, h.best_t, h.hit_n, h.hit_prim, h.phantom_t, h.runner_t, c.tris, c.analytic, excl, pb + 2u * i);
    }
    for (uint32_t i = 0; i < sv.pro_spheres; ++i)
        test_prim<COUNTERS, false, false, TABS>(sv, PRIM_SPHERE, i, org, dir, inv_d, h.best_t, h.hit_n, h.hit_prim, h.phantom_t, h.runner_t, c.tris, c.analytic, excl, ps + i);
    for (uint32_t i = 0; i < sv.pro_cyls; ++i)
        test_prim<COUNTERS, false, false, TABS>(sv, PRIM_CYL, i, org, dir, inv_d, h.best_t, h.hit_n, h.hit_prim, h.phantom_t, h.runner_t, c.tris, c.analytic, excl, pc + 4u * i);
}

#ifndef ORT_HOST_SIM
/* ---- the same walk, by the wave -------------------------------------------------------------------------------
 * One lane's ray (lane `leader`), walked by all the lanes that are active with it (the others of resolve_hit's callers:
 * any subset, at least the leader).  The order-dependent part of raycast_bvh is "best at that moment": a node's records
 * are tested against the best so far, its children are admitted against the best after its records, nodes are taken in
 * queue order.  Nodes are still taken one at a time, in order; what the lanes share is the work inside a node:
 *   records: one per lane, each against the best so far; the sequential rule (take when t < best, strictly) ends with
 *            the smallest t, the earliest record among equals -- picked here from the lanes that would take theirs;
 *   children: one per lane, all eight against the same best (nothing changes it between them); the admitted ones are
 *            appended in slot order by the leader lane, which is also the only lane that reads the queue (program
 *            order of ONE thread keeps its stores and loads of the queue coherent).
 * A node then costs about four dependent round trips instead of one per record and per child: on the 1M-triangle
 * scene the walks took 10 % of the launch (one lane walking, 63 waiting) before this. */
ORT_D float wave_bcast(float v, int lane) { return om_bits_f32((uint32_t)__builtin_amdgcn_readlane((int)om_f32_bits(v), lane)); }
ORT_D uint32_t wave_bcast(uint32_t v, int lane) { return (uint32_t)__builtin_amdgcn_readlane((int)v, lane); }
ORT_D uint32_t rank_in(unsigned long long mask) { /* set bits of mask below this lane */
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}

template <bool COUNTERS>
ORT_D bool ref_raycast_bfs_wave(const SceneView &sv, const int leader, V3 org_l, V3 dir_l, V3 inv_l, uint32_t *queue, float &out_t, V3 &out_n,
                                uint32_t &out_prim, unsigned long long &c_nodes, unsigned long long &c_tris, unsigned long long &c_analytic) {
    /* the leader's ray, in VECTOR registers of every lane (the empty asm hides their uniformity: as scalars they would
       crowd the scalar register file, whose spills land in the hot code around this rare region) */
    V3 org = mk(wave_bcast(org_l.x, leader), wave_bcast(org_l.y, leader), wave_bcast(org_l.z, leader));
    V3 dir = mk(wave_bcast(dir_l.x, leader), wave_bcast(dir_l.y, leader), wave_bcast(dir_l.z, leader));
    V3 inv_d = mk(wave_bcast(inv_l.x, leader), wave_bcast(inv_l.y, leader), wave_bcast(inv_l.z, leader));
    asm volatile("" : "+v"(org.x), "+v"(org.y), "+v"(org.z), "+v"(dir.x), "+v"(dir.y), "+v"(dir.z), "+v"(inv_d.x), "+v"(inv_d.y), "+v"(inv_d.z));
    const unsigned long long active = __ballot(true);
    const uint32_t n_active = (uint32_t)__popcll(active), rank = rank_in(active);
    const bool is_leader = ORT_LANE() == leader;
    /* wave-uniform state of the walk */
    float best_t = 3.402823466e+38f;
    int win_lane = -1;            /* the lane that holds the best hit's normal and primitive (keep_*) */
    V3 keep_n = mk(0, 0, 0);
    uint32_t keep_prim = kNoPrim;
    uint32_t head = 0, tail = 1;
    bool ok = true;
    const uint32_t cap = sv.cold->bfs_queue_cap;
    if (is_leader) queue[0] = 0;
    while (head != tail) {
        uint32_t node_v = 0;
        if (is_leader) node_v = queue[head];
        const uint32_t node = wave_bcast(node_v, leader);
        head++;
        const float4 *np = sv.cold->ref_nodes + 3u * node;
        const float4 a = np[0], b = np[1], c = np[2];
        const int32_t first_child = (int32_t)om_f32_bits(a.w);
        const uint32_t rec_first = om_f32_bits(b.w), rec_count = om_f32_bits(c.x);
        for (uint32_t r0 = 0; r0 < rec_count; r0 += n_active) {
            float t = best_t, unused = 0;
            V3 n = mk(0, 0, 0);
            uint32_t prim = kNoPrim;
            if (r0 + rank < rec_count) {
                const uint32_t rec = sv.cold->ref_recs[rec_first + r0 + rank];
                test_prim<COUNTERS, true>(sv, rec >> 28, rec & 0x00ffffffu, org, dir, inv_d, t, n, prim, unused, unused, c_tris, c_analytic);
            }
            /* the lanes that would take their record: the smallest distance wins, the lowest lane (= earliest record) among equals */
            unsigned long long takers = __ballot(prim != kNoPrim);
            if (takers != 0ull) {
                int w = -1;
                float m = 0.0f;
                while (takers != 0ull) {
                    const int j = __ffsll(takers) - 1;
                    takers &= takers - 1ull;
                    const float tj = wave_bcast(t, j);
                    if (w < 0 || tj < m) { m = tj; w = j; }
                }
                best_t = m;
                win_lane = w;
                if (ORT_LANE() == w) { keep_n = n; keep_prim = prim; }
            }
        }
        if (first_child >= 0) {
            for (uint32_t k0 = 0; k0 < 8u; k0 += n_active) {
                const uint32_t k = k0 + rank;
                bool add = false;
                if (k < 8u) {
                    const float4 *cp = sv.cold->ref_nodes + 3u * ((uint32_t)first_child + k);
                    const float4 ca = cp[0], cb = cp[1], cc = cp[2];
                    const uint32_t flags = om_f32_bits(cc.y);
                    const bool leaf_with_records = (flags & 3u) == 3u;
                    const bool has_children = (int32_t)om_f32_bits(ca.w) >= 0;
                    if (leaf_with_records || has_children) {
                        const V3 lo = mk(ca.x, ca.y, ca.z), hi = mk(cb.x, cb.y, cb.z);
                        add = (org.x >= lo.x && org.x < hi.x) && (org.y >= lo.y && org.y < hi.y) && (org.z >= lo.z && org.z < hi.z);
                        if (!add) {
                            const float t = hit_aab_t(lo, hi, org, inv_d);
                            if (COUNTERS) c_nodes++;
                            add = (t >= kHitTMin && t < best_t);
                        }
                    }
                }
                const unsigned long long admitted = __ballot(add);
                const uint32_t n_add = (uint32_t)__popcll(admitted);
                if (is_leader) { /* in lane order = slot order */
                    unsigned long long m2 = admitted;
                    uint32_t at = tail;
                    while (m2 != 0ull) {
                        const int j = __ffsll(m2) - 1;
                        m2 &= m2 - 1ull;
                        const uint32_t kk = k0 + (uint32_t)__popcll(active & ((1ull << j) - 1ull));
                        if (at < cap) queue[at++] = (uint32_t)first_child + kk;
                    }
                }
                if (tail + n_add > cap) { ok = false; tail = cap; } else tail += n_add;
            }
        }
    }
    V3 hit_n = mk(0, 0, 0);
    uint32_t hit_prim = kNoPrim;
    if (win_lane >= 0) {
        hit_n = mk(wave_bcast(keep_n.x, win_lane), wave_bcast(keep_n.y, win_lane), wave_bcast(keep_n.z, win_lane));
        hit_prim = wave_bcast(keep_prim, win_lane);
    }
    if (is_leader) {
        ORT_COUNT(sv.cold->fallback_counters + kDiagFallback, (unsigned long long)tail);
        out_t = best_t; out_n = hit_n; out_prim = hit_prim;
    }
    return ok;
}
#endif /* !ORT_HOST_SIM */

/* the exact answer: raycast_bvh emulated literally on the reference-compatible octree.  Rare.  The lanes of a
   wave that need it take turns (wave-uniform loop over the ballot), so a wave never has more than one lane
   holding or waiting for a queue of the pool: a waiting lane can only wait for holders in other waves, which
   are running, never for a lane of its own wave parked at a reconvergence point.  The fences order the queue's
   contents across holders on different XCDs (each XCD has its own L2). */
template <bool COUNTERS>
ORT_D void recast_exactly(const SceneView &sv, bool need, V3 org, V3 dir, V3 inv_d, uint32_t lane_id, HitState &h, Counters &c) {
#ifdef ORT_MEASURE_NO_RECAST /* developer measurement only (wrong images): what the exact fallback costs, by leaving it out */
    return;
#endif
    unsigned long long pending = ORT_BALLOT(need);
    while (ORT_RARE(pending != 0ull)) {
        const int leader = ORT_FFS64(pending) - 1;
        uint32_t slot = 0;
        if (ORT_LANE() == leader) {
            ORT_COUNT(sv.cold->fallback_counters, 1ull); /* straight to memory, no register kept across the loop */
            /* a queue of the pool: look before trying (a plain load does not serialise in L2 the way an atomic on a
               contended line does) and back off between looks.  Without the back-off the lanes that wait slow the
               breadth-first walks that hold the queues down (every load of theirs queues up behind the atomics), which
               makes more lanes wait: on a long launch over a 1M-triangle scene that feedback halved the throughput */
            slot = ((lane_id * 2654435761u) >> 8) % sv.cold->bfs_queue_count;
            while (ORT_PEEK(sv.cold->bfs_locks + slot * kBfsLockStride) != 0u || !ORT_TRY_LOCK(sv.cold->bfs_locks + slot * kBfsLockStride)) {
                ORT_COUNT(sv.cold->fallback_counters + kDiagFallback + 2, 1ull); /* diagnostics (ORT_DEBUG_FALLBACK): busy queues met */
                slot = (slot + 1u) % sv.cold->bfs_queue_count;
                ORT_BACKOFF();
            }
        }
        ORT_FENCE();
#ifdef ORT_HOST_SIM
        const bool ok = ref_raycast_bfs<COUNTERS>(sv, org, dir, inv_d, sv.cold->bfs_pool + (size_t)slot * sv.cold->bfs_queue_cap, h.best_t, h.hit_n,
                                                  h.hit_prim, c.nodes, c.tris, c.analytic);
#else
        /* the lanes that are here with the leader walk its ray together (ref_raycast_bfs_wave) */
        slot = wave_bcast(slot, leader);
        const bool ok = ref_raycast_bfs_wave<COUNTERS>(sv, leader, org, dir, inv_d, sv.cold->bfs_pool + (size_t)slot * sv.cold->bfs_queue_cap, h.best_t, h.hit_n,
                                                       h.hit_prim, c.nodes, c.tris, c.analytic);
#endif
        ORT_FENCE();
        if (ORT_LANE() == leader) {
            if (!ok) ORT_COUNT(sv.cold->fallback_counters + 1, 1ull);
            ORT_UNLOCK(sv.cold->bfs_locks + slot * kBfsLockStride);
        }
        pending &= pending - 1ull;
    }
}

/* ray.cpp:1215-1221: the point on the focal plane through the centre of pixel pxy = x | y << 16 */
ORT_D V3 focal_point(const RenderHot &rv, uint32_t pxy, V3 cam_p, V3 cam_x, V3 cam_y, V3 cam_z, float focal_length) {
    float fx = (2.0f * (int)(pxy & 0xffffu) / (float)rv.W) - 1.0f; /* i32 -> f32, as the reference's x, y */
    float fy = (2.0f * (int)(pxy >> 16) / (float)rv.H) - 1.0f;
    V3 to_pixel = normalize(sub(add(scale(fx, cam_x), scale(fy, cam_y)), cam_z));
    return add(cam_p, scale(focal_length, to_pixel));
}

/* Advance the lane's path state machine until it has produced the next ray (returns true; the ray
   is P.org / P.dir) or has run out of work (returns false).  On entry with P.ps == PS_HIT, h holds
   the resolved closest hit of the ray produced by the previous call. */
#ifndef ORT_HOST_SIM
/* Job indices for the lanes of this wave that need one now (the active lanes).  The wave draws them from next_job in
   batches -- one returned device-scope atomic, a few microseconds that the whole wave waits for, per job_batch jobs
   instead of one in every pass in which some lane ends a job -- and keeps the unissued part of its last batch in two
   words of LDS: pool[0] = next index, pool[1] = end of the batch.  What the batches are really for: consecutive
   indices are like jobs (block-major order: the 64 pixels of one 8x8 block under one chunk seed), so the lanes of a
   wave trace like paths and end their jobs together -- plain loop on the headline frame 4 728 -> 5 204 Mpaths/s,
   analytic scene 3 567 -> 3 886 (profiles/r03_tuning.md).  Near the end of the job space (batch_until) a wave asks
   for exactly what it needs, so that no wave sits on indices that idle lanes elsewhere could take.  Which lane runs
   which job cannot change a bit of the image (seeds belong to jobs). */
ORT_D unsigned long long draw_job(const RenderHot &rv, unsigned long long *pool) {
    const unsigned long long mask = __ballot(true);
    const uint32_t n = (uint32_t)__popcll(mask);
    const uint32_t rank = (uint32_t)__builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
    const unsigned long long next = ((volatile unsigned long long *)pool)[0], end = ((volatile unsigned long long *)pool)[1];
    const uint32_t avail = (uint32_t)(end - next);
    unsigned long long j = next + rank;
    if (n > avail) { /* wave-uniform */
        const uint32_t need = n - avail;
        uint32_t want = rv.c->job_batch;
        if (want < need || end >= rv.c->batch_until) want = need;
        unsigned long long base = 0ull;
        if (rank == 0u) base = atomicAdd(rv.c->next_job, (unsigned long long)want);
        const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)base);
        const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(base >> 32));
        base = ((unsigned long long)hi << 32) | lo;
        if (rank >= avail) j = base + (rank - avail);
        if (rank == 0u) { ((volatile unsigned long long *)pool)[0] = base + need; ((volatile unsigned long long *)pool)[1] = base + want; }
    } else if (rank == 0u) {
        ((volatile unsigned long long *)pool)[0] = next + n;
    }
    return j;
}
/* this wave's two words of the workgroup's pool array, emptied; nullptr when draws are not batched */
ORT_D unsigned long long *wave_job_pool(const RenderHot &rv, unsigned long long *lds_pool) {
    unsigned long long *pool = lds_pool + 2u * (threadIdx.x >> 6);
    if ((threadIdx.x & 63u) == 0u) { ((volatile unsigned long long *)pool)[0] = 0ull; ((volatile unsigned long long *)pool)[1] = 0ull; }
    return rv.c->job_batch > 1u ? pool : nullptr;
}
#endif

/* IMPLICIT: the caller vouches for an implicit job space (PIXEL / CHUNK policies: every job is one pixel, spp_u
   samples): the job's rect, its sample count and its index then need no registers of their own */
template <bool COUNTERS, bool DIFFUSE = false, bool TABS = false, bool IMPLICIT = false>
ORT_D bool produce_ray(const SceneView &sv, const RenderHot &rv, const float4 *tab, PathState &P, const HitState &h, Counters &c, Prof &pr,
                       float *focal_cache = nullptr, int focal_stride = 0, uint32_t spp_u = 0, uint32_t *late_flag = nullptr, bool no_new_job = false,
                       unsigned long long *pool = nullptr) {
    const V3 cam_p = mk(sv.cam[0], sv.cam[1], sv.cam[2]);
    const V3 cam_x = mk(sv.cam[3], sv.cam[4], sv.cam[5]);
    const V3 cam_y = mk(sv.cam[6], sv.cam[7], sv.cam[8]);
    const V3 cam_z = mk(sv.cam[9], sv.cam[10], sv.cam[11]);
    const float focal_length = len(sub(cam_p, mk(0, 0, 0.2f))); /* ray.cpp:1198 */
    const float aperture = 0.1f;                                /* ray.cpp:1199 */

    while (P.ps != PS_DONE) {
        ORT_UTIL(sv, 5, true);
        ORT_PHASE(pr, sv, 8, true);
        bool bounce = false;
        float angle = 0.0f;
        BrdfDraw draw;
        Mat m;
        V3 n, focal;
        if (P.ps == PS_HIT) {
            /* a traversal has finished: ray.cpp:817 then :1251-1277 (primary) or :1355-1421 (bounce) */
            bool alive = true;
            const uint32_t hit_mat = h.hit_mat; /* resolve_hit: 0 = nothing hit */
            n = normalize(h.hit_n);
            ORT_SIM_RAY_HOOK((int)(P.pxy & 0xffffu), (int)(P.pxy >> 16), P.org, P.dir, h.best_t, n, hit_mat);
            if (COUNTERS && P.primary) c.paths++;
            if (hit_mat) {
                if (TABS) m = load_mat(tab + kTabMats, hit_mat);
                else m = load_mat(sv.materials, hit_mat);
            }
            if (!hit_mat) {
                alive = false; /* bounce miss: ray.cpp:1418-1421; primary miss: undefined in the reference, defined: terminate */
            } else if (m.is_light) {
                /* ray.cpp:1254-1259 (primary: unweighted, unchecked) / :1358-1371 (bounce: dropped if not finite) */
                V3 e = P.primary ? m.emit : had(P.weight, m.emit);
                if (P.primary || (!isnan3(e) && !isinf3(e))) P.color = add(P.color, e);
                alive = false;
            } else {
                if (P.primary) {
                    if (len2(m.kd) > 0.0f) P.weight = had(P.weight, m.kd); /* ray.cpp:1267-1270 */
                } else {
                    /* ray.cpp:1374-1405: pdf and BSDF with the NEW surface's normal and material, the OLD wo (sic) */
                    if (DIFFUSE) {
                        float p = pdf_brdf<true>(n, P.dir, P.wo, kRoughness, m) * rv.rr;
                        if (p > 0.000001f) {
                            V3 f = eval_scattering<true>(n, P.dir, P.wo, m, kRoughness, h.best_t);
                            P.weight = had(divs(f, p), P.weight);
                        }
                    } else { /* all lobes: the two calls fused, their common terms evaluated once (ort_device.h) */
                        V3 f = mk(0, 0, 0);
                        const float p = pdf_eval_scattering(n, P.dir, P.wo, m, kRoughness, h.best_t, rv.rr, f);
                        if (p > 0.000001f) P.weight = had(divs(f, p), P.weight);
                    }
                    P.wo = neg(P.dir);
                }
                P.org = add(P.org, scale(h.best_t - kEps, P.dir)); /* ray.cpp:1262,1411 */
            }
            P.primary = false;
            /* ray.cpp:1280: the roulette draw happens only while the path is alive */
            bounce = alive && rng_01(P.rng) < rv.rr;
            if (bounce) {
                /* sample_random_lights (ray.cpp:537-601): result unused, RNG advances */
                rng_step(P.rng);
                if (sv.light_count) {
                    uint32_t li = P.rng % sv.light_count;
                    uint32_t is_sphere;
                    if (TABS) is_sphere = ((const uint32_t *)tab)[4 * kTabLights + li];
                    else is_sphere = sv.light_is_sphere[li];
                    if (is_sphere) { rng_step(P.rng); rng_step(P.rng); rng_step(P.rng); rng_step(P.rng); }
                }
                ORT_UTIL(sv, 6, true);
                draw = sample_brdf_draw<DIFFUSE>(P.rng, kRoughness, m);
                angle = draw.phi;
            } else {
                P.sample++;
                P.ps = PS_SAMPLE;
            }
            ORT_PHASE(pr, sv, 1, true);
        }
        if (!bounce) {
            /* a lane arrives here after its sample ended (PS_SAMPLE), or with nothing yet (PS_NEED_JOB).
               Pixel write-back, next pixel / next job and the new camera ray all happen in this same
               pass, so the rest of the wave does not wait through a second trip round the loop. */
            const uint32_t job_spp = IMPLICIT ? spp_u : P.spp;
            if (P.ps == PS_SAMPLE && P.sample == job_spp) {
                /* ray.cpp:1428 */
                V3 o = divs(P.color, (float)job_spp);
                uint32_t px = P.pxy & 0xffffu, py = P.pxy >> 16;
                float *p = pixel_ptr(rv, P.jyp >> 16, px, py);
                p[0] = o.x; p[1] = o.y; p[2] = o.z;
                if (IMPLICIT) {
                    P.ps = PS_NEED_JOB; /* a one-pixel job ends with its pixel */
                } else {
                px++;
                if (px == (P.jxx >> 16)) { px = P.jxx & 0xffffu; py++; }
                P.pxy = px | (py << 16);
                if (py == (P.jyp & 0xffffu)) {
                    if (rv.mode == JOBS_EXPLICIT && rv.c->final_states) rv.c->final_states[P.job_index] = P.rng;
                    P.ps = PS_NEED_JOB;
                } else {
                    P.ps = PS_PIXEL;
                }
                }
            }
            if (P.ps == PS_NEED_JOB) {
                if (no_new_job) return false; /* ray exchange, end of the launch: this lane takes a parked path first (pt_lane_x) */
                unsigned long long j;
#ifndef ORT_HOST_SIM
                if (pool) j = draw_job(rv, pool);
                else
#endif
                j = ORT_NEXT_JOB(rv.c->next_job);
                if (j >= rv.c->job_count) { P.ps = PS_DONE; break; }
                if (IMPLICIT && late_flag && j >= rv.c->endgame_from) *late_flag = 1u; /* ray exchange: the launch is near its end (pt_lane_x) */
                if (!IMPLICIT && rv.mode == JOBS_EXPLICIT) {
                    ort_tile_job jb = rv.c->jobs[j];
                    P.job_index = (uint32_t)j;
                    P.jxx = (uint32_t)jb.x0 | ((uint32_t)jb.x1 << 16);
                    P.jyp = (uint32_t)jb.y1;
                    P.pxy = (uint32_t)jb.x0 | ((uint32_t)jb.y0 << 16);
                    P.rng = jb.rng_state; P.spp = jb.spp;
                    if (jb.x1 <= jb.x0 || jb.y1 <= jb.y0) { /* empty rect: the reference loops zero times */
                        if (rv.c->final_states) rv.c->final_states[P.job_index] = P.rng;
                        continue;
                    }
                } else {
                    /* implicit job space: [chunk k][my 8x8 block b][pixel-in-block p] */
                    unsigned long long per_chunk = (unsigned long long)rv.c->my_blocks * 64ull;
                    uint32_t k, lb, pin; /* chunk, local block, pixel in block */
                    if (rv.c->block_major) {
                        const uint32_t per_block = rv.c->nchunks * 64u;
                        const uint32_t within = (uint32_t)(j % per_block);
                        lb = (uint32_t)(j / per_block);
                        k = within >> 6;
                        pin = within & 63u;
                    } else {
                        k = (uint32_t)(j / per_chunk);
                        const uint32_t rem = (uint32_t)(j % per_chunk);
                        lb = rem >> 6;
                        pin = rem & 63u;
                    }
                    uint32_t blk = rv.c->shard_index + lb * rv.c->shard_count;
                    int x = (int)((rv.c->block_x0 + blk % rv.c->blocks_w) * 8u + (pin & 7u));
                    int y = (int)((rv.c->block_y0 + blk / rv.c->blocks_w) * 8u + (pin >> 3));
                    if (x < rv.c->x0 || x >= rv.c->x1 || y < rv.c->y0 || y >= rv.c->y1) continue;
                    uint32_t pix = (uint32_t)(y * rv.W + x);
                    if (!IMPLICIT) P.jxx = (uint32_t)x | ((uint32_t)(x + 1) << 16);
                    P.pxy = (uint32_t)x | ((uint32_t)y << 16);
                    if (rv.mode == JOBS_PIXEL) {
                        P.rng = job_seed(rv.c->seed, pix);
                        if (!IMPLICIT) P.spp = rv.c->spp;
                        P.jyp = (uint32_t)(y + 1);
                    } else {
                        P.rng = job_seed(rv.c->seed, k * (uint32_t)(rv.W * rv.H) + pix);
                        if (!IMPLICIT) P.spp = rv.c->chunk;
                        P.jyp = (uint32_t)(y + 1) | (k << 16);
                    }
                }
                P.ps = PS_PIXEL;
            }
            if (P.ps == PS_PIXEL) {
                ORT_SIM_PIXEL_HOOK((int)(P.pxy & 0xffffu), (int)(P.pxy >> 16), P.rng);
                P.color = mk(0, 0, 0); /* ray.cpp:1211 */
                P.sample = 0;
                P.ps = PS_SAMPLE;
                if (focal_cache) { /* the pixel's focal point, once per pixel (persistent kernel: three floats of LDS per lane) */
                    V3 f = focal_point(rv, P.pxy, cam_p, cam_x, cam_y, cam_z, focal_length);
                    focal_cache[0] = f.x; focal_cache[focal_stride] = f.y; focal_cache[2 * focal_stride] = f.z;
                }
            }
            if (P.sample == job_spp) continue; /* spp == 0: the reference's sample loop runs zero times */
            /* ray.cpp:1215-1221: point on the focal plane through the pixel centre: a function of the pixel alone,
               read back from the per-lane cache or (wavefront mode) recomputed -- same expressions, same bits */
            focal = focal_cache ? mk(focal_cache[0], focal_cache[focal_stride], focal_cache[2 * focal_stride])
                                : focal_point(rv, P.pxy, cam_p, cam_x, cam_y, cam_z, focal_length);
            angle = rng_between(P.rng, 0.0f, 2 * kPi); /* ray.cpp:1232 */
            ORT_PHASE(pr, sv, 2, true);
        }
        /* lanes that bounce and lanes that start a new camera sample both need cos/sin of one angle
           (lobe azimuth / aperture angle): the double-precision evaluation happens here once,
           converged, instead of once in each branch (same operand, same bits) */
        ORT_UTIL(sv, 7, true);
        float cs, sn;
        ort_sincosf(angle, &sn, &cs);
        if (DIFFUSE) { /* the leaner flavour has the registers for the wider merge; the all-lobes one spills on it */
            /* Bounce lanes normalise twice here (the surface normal again, ray.cpp:1069, and the sampled direction,
               :1158) and so do camera lanes (the ray direction, :1240, and -- sic -- the direction again for wo,
               :1241): two converged evaluations instead of four divergent ones */
            V3 ap = mk(0, 0, 0);
            if (!bounce) /* ray.cpp:1233-1239 */
                ap = sub(add(add(cam_p, scale(aperture * cs, cam_x)), scale(aperture * sn, cam_y)), scale(0.1f, cam_z));
            const V3 unit1 = normalize(bounce ? n : sub(focal, ap));
            bool is_trans = false;
            V3 raw = unit1;
            if (bounce) raw = sample_brdf_finish<false, true>(n, unit1, P.wo, m, draw, cs, sn, is_trans);
            const V3 unit2 = normalize(raw);
            if (bounce) {
                if (is_trans) P.org = add(P.org, scale(2.0f * kEps, P.dir)); /* ray.cpp:1345-1348: dir is still the arriving direction */
                P.dir = unit2;
            } else {
                P.dir = unit1;
                P.wo = neg(unit2);
                P.org = ap;
                P.weight = mk(1, 1, 1);
                P.primary = true;
                P.ps = PS_HIT;
            }
        } else {
            /* both branches end by normalising a direction (ray.cpp:1158 / :1240): one converged evaluation */
            V3 raw, ap = mk(0, 0, 0);
            bool is_trans = false;
            if (bounce) {
                raw = sample_brdf_finish<false>(n, normalize(n), P.wo, m, draw, cs, sn, is_trans);
            } else {
                /* ray.cpp:1233-1246 */
                ap = sub(add(add(cam_p, scale(aperture * cs, cam_x)), scale(aperture * sn, cam_y)), scale(0.1f, cam_z));
                raw = sub(focal, ap);
            }
            const V3 unit = normalize(raw);
            if (bounce) {
                if (is_trans) P.org = add(P.org, scale(2.0f * kEps, P.dir)); /* ray.cpp:1345-1348: dir is still the arriving direction */
                P.dir = unit;
            } else {
                P.dir = unit;
                P.wo = neg(normalize(P.dir)); /* normalised again (sic) */
                P.org = ap;
                P.weight = mk(1, 1, 1);
                P.primary = true;
                P.ps = PS_HIT;
            }
        }
        ORT_PHASE(pr, sv, 3, true);
        return true;
    }
    return false;
}


/* a child is skipped only when its entry distance (less the slab test's rounding margin, 0.9999996)
   is beyond best_t * 1.0002: every primitive hit within 2e-4 of the final winner is therefore tested,
   which is what makes HitState::runner_t exact in that window */
constexpr float kCullSlack = 0.9997996f; /* <= 0.9999996 / 1.0002 */

/* One interior node of the fast tree (record a b cc d) against the ray: both child boxes in the reference's own
   (p - o) * (1/d) form (ray.cpp:215-222) with ulp margins, conservative; fminf/fmaxf drop the NaN of 0 * inf, i.e. that
   axis is ignored.  Continues with the nearer child, stacks the farther one, or pops. */
template <bool COUNTERS, int LDS_ENTRIES, int BLOCK>
ORT_D void visit_node(float4 a, float4 b, float4 cc, float4 d, V3 org, V3 inv_d, float best_t, uint32_t &cur, int &sp, uint32_t *lds_stack,
                      uint32_t *spill, int tid, Counters &c) {
    uint32_t c0 = om_f32_bits(d.x), c1 = om_f32_bits(d.y);
    if (COUNTERS) c.nodes += 2;
    /* child 0: lo = a.xyz, hi = (a.w, b.x, b.y); child 1: lo = (b.z, b.w, cc.x), hi = cc.yzw */
    float t0x = (a.x - org.x) * inv_d.x, t1x = (a.w - org.x) * inv_d.x;
    float t0y = (a.y - org.y) * inv_d.y, t1y = (b.x - org.y) * inv_d.y;
    float t0z = (a.z - org.z) * inv_d.z, t1z = (b.y - org.z) * inv_d.z;
    float n0 = fmaxf(fmaxf(fminf(t0x, t1x), fminf(t0y, t1y)), fminf(t0z, t1z));
    float f0 = fminf(fminf(fmaxf(t0x, t1x), fmaxf(t0y, t1y)), fmaxf(t0z, t1z));
    float u0x = (b.z - org.x) * inv_d.x, u1x = (cc.y - org.x) * inv_d.x;
    float u0y = (b.w - org.y) * inv_d.y, u1y = (cc.z - org.y) * inv_d.y;
    float u0z = (cc.x - org.z) * inv_d.z, u1z = (cc.w - org.z) * inv_d.z;
    float n1 = fmaxf(fmaxf(fminf(u0x, u1x), fminf(u0y, u1y)), fminf(u0z, u1z));
    float f1 = fminf(fminf(fmaxf(u0x, u1x), fmaxf(u0y, u1y)), fmaxf(u0z, u1z));
    /* children with a sphere below are not culled by distance (phantom tangent hits) */
    bool h0 = (f0 * 1.0000004f >= n0) && (f0 >= 0.0f) && ((n0 * kCullSlack < best_t) || (c0 & SPHERE_BELOW_BIT));
    bool h1 = (f1 * 1.0000004f >= n1) && (f1 >= 0.0f) && ((n1 * kCullSlack < best_t) || (c1 & SPHERE_BELOW_BIT)) && (c1 != EMPTY_CHILD);
    if (h0 && h1) {
        bool swap = n1 < n0;
        uint32_t farc = swap ? c0 : c1;
        cur = swap ? c1 : c0;
        if (sp < LDS_ENTRIES) lds_stack[sp * BLOCK + tid] = farc;
        else spill[sp - LDS_ENTRIES] = farc;
        sp++;
    } else if (h0) {
        cur = c0;
    } else if (h1) {
        cur = c1;
    } else if (sp == 0) {
        cur = kTraversalDone;
    } else {
        sp--;
        /* two real loads behind a branch (the volatile keeps the compiler from merging them into one
           flat_load of a selected generic pointer): the LDS side becomes a plain ds_read */
        if (sp < LDS_ENTRIES) cur = lds_stack[sp * BLOCK + tid];
        else cur = ((volatile uint32_t *)spill)[sp - LDS_ENTRIES];
    }
}

/* One interior node of the 4-wide tree (DevNode4: lo.x, lo.y, lo.z, hi.x, hi.y, hi.z of four children, four child words)
   against the ray: the same conservative slab test as visit_node per child, then the children that are hit are visited
   nearest first -- the nearest becomes the current node, the others are stacked farthest first.  Half the dependent
   fetches of the binary tree on the way down. */
template <bool COUNTERS, int LDS_ENTRIES, int BLOCK>
ORT_D void visit_node4(float4 lx, float4 ly, float4 lz, float4 hx, float4 hy, float4 hz, float4 cw, V3 org, V3 inv_d, float best_t, uint32_t &cur, int &sp,
                       uint32_t *lds_stack, uint32_t *spill, int tid, Counters &c) {
    if (COUNTERS) c.nodes += 4;
    float key0, key1, key2, key3;
    uint32_t w0, w1, w2, w3;
#define ORT_SLAB4(K, W, LX, LY, LZ, HX, HY, HZ, CW)                                                                        \
    {                                                                                                                      \
        const float t0x = ((LX) - org.x) * inv_d.x, t1x = ((HX) - org.x) * inv_d.x;                                        \
        const float t0y = ((LY) - org.y) * inv_d.y, t1y = ((HY) - org.y) * inv_d.y;                                        \
        const float t0z = ((LZ) - org.z) * inv_d.z, t1z = ((HZ) - org.z) * inv_d.z;                                        \
        const float n_ = fmaxf(fmaxf(fminf(t0x, t1x), fminf(t0y, t1y)), fminf(t0z, t1z));                                  \
        const float f_ = fminf(fminf(fmaxf(t0x, t1x), fmaxf(t0y, t1y)), fmaxf(t0z, t1z));                                  \
        const uint32_t cw_ = om_f32_bits(CW);                                                                              \
        const bool hit_ = (f_ * 1.0000004f >= n_) && (f_ >= 0.0f) && ((n_ * kCullSlack < best_t) || (cw_ & SPHERE_BELOW_BIT)) && (cw_ != EMPTY_CHILD); \
        K = hit_ ? fminf(n_, 3.402823466e+38f) : __builtin_inff(); /* a hit sorts strictly before every miss */             \
        W = hit_ ? cw_ : EMPTY_CHILD;                                                                                      \
    }
    ORT_SLAB4(key0, w0, lx.x, ly.x, lz.x, hx.x, hy.x, hz.x, cw.x)
    ORT_SLAB4(key1, w1, lx.y, ly.y, lz.y, hx.y, hy.y, hz.y, cw.y)
    ORT_SLAB4(key2, w2, lx.z, ly.z, lz.z, hx.z, hy.z, hz.z, cw.z)
    ORT_SLAB4(key3, w3, lx.w, ly.w, lz.w, hx.w, hy.w, hz.w, cw.w)
#undef ORT_SLAB4
    /* ascending by entry distance: five compare-exchanges */
#define ORT_CX(KA, WA, KB, WB)                                                                                             \
    {                                                                                                                      \
        const bool sw_ = KB < KA;                                                                                          \
        const float ka_ = sw_ ? KB : KA, kb_ = sw_ ? KA : KB;                                                              \
        const uint32_t wa_ = sw_ ? WB : WA, wb_ = sw_ ? WA : WB;                                                           \
        KA = ka_; KB = kb_; WA = wa_; WB = wb_;                                                                            \
    }
    ORT_CX(key0, w0, key1, w1)
    ORT_CX(key2, w2, key3, w3)
    ORT_CX(key0, w0, key2, w2)
    ORT_CX(key1, w1, key3, w3)
    ORT_CX(key1, w1, key2, w2)
#undef ORT_CX
    if (w0 == EMPTY_CHILD) { /* nothing hit: pop */
        if (sp == 0) {
            cur = kTraversalDone;
        } else {
            sp--;
            if (sp < LDS_ENTRIES) cur = lds_stack[sp * BLOCK + tid];
            else cur = ((volatile uint32_t *)spill)[sp - LDS_ENTRIES];
        }
        return;
    }
    cur = w0;
    if (w3 != EMPTY_CHILD) { if (sp < LDS_ENTRIES) lds_stack[sp * BLOCK + tid] = w3; else spill[sp - LDS_ENTRIES] = w3; sp++; }
    if (w2 != EMPTY_CHILD) { if (sp < LDS_ENTRIES) lds_stack[sp * BLOCK + tid] = w2; else spill[sp - LDS_ENTRIES] = w2; sp++; }
    if (w1 != EMPTY_CHILD) { if (sp < LDS_ENTRIES) lds_stack[sp * BLOCK + tid] = w1; else spill[sp - LDS_ENTRIES] = w1; sp++; }
}

/* raycast_top_most_node (ray.cpp:1165-1176): start at the root.  The lanes that start a ray now are converged: they
   test the analytic prologue together and visit the root node together, its record read from the LDS tables, before
   they join the traversal loop (whose other lanes are at arbitrary depths) */
template <bool COUNTERS, bool TABS, int LDS_ENTRIES, int BLOCK>
ORT_D void begin_ray(const SceneView &sv, const float4 *tab, const PathState &P, Trav &T, HitState &h, Counters &c, Prof &pr,
                     uint32_t *lds_stack, uint32_t *spill, int tid) {
    T.cur = 0;
    T.sp = 0;
    T.inv_d = mk(1.0f / P.dir.x, 1.0f / P.dir.y, 1.0f / P.dir.z); /* ray.cpp:210, once per ray */
    reset_hit(h, 3.402823466e+38f); /* Flt_Max, ray.cpp:627 */
    prologue_tests<COUNTERS, TABS>(sv, tab, P.org, P.dir, T.inv_d, h, c);
    ORT_PHASE(pr, sv, 4, true);
}

/* Closest hit: interruptible ordered DFS, replaces raycast_bvh (ray.cpp:624-822) on the fast tree.
 * while-while: a lane first descends interior nodes until it holds a leaf (or runs out of stack),
 * then the wave processes leaves together, so the cheap box code and the expensive primitive code
 * are not serialised against each other in every iteration.
 * Slab test: the reference's own (p - o) * (1/d) form (ray.cpp:215-222) with ulp margins,
 * conservative; fminf/fmaxf drop the NaN of 0 * inf, i.e. that axis is ignored.
 * Returns when this lane's ray is finished, or -- refill_below > 0 -- as soon as fewer than
 * refill_below lanes of the wave are still traversing (the caller resumes later: all state is in T/h). */
template <bool COUNTERS, int LDS_ENTRIES, int BLOCK, bool TREELET = false, bool ANNOUNCE = false, bool WIDE = false>
ORT_D bool traverse(const SceneView &sv, V3 org, V3 dir, Trav &T, HitState &h, uint32_t *lds_stack, uint32_t *spill, int tid,
                    int refill_below, int descend_below, Counters &c, Prof &pr, uint32_t excl = kNoPrim, const float4 *tab = nullptr) {
    bool tracing = true;
    uint32_t cur = T.cur;
    int sp = T.sp;
    const V3 inv_d = T.inv_d;
    while (tracing) {
        ORT_UTIL(sv, 2, true);
        ORT_PHASE(pr, sv, 9, true);
        /* the straggler threshold of this round: descend_below, but never more than a quarter of the lanes
           that start descending now (a wave that enters with 20 such lanes should not stop at 8) */
        const int entering = ORT_POPC64(ORT_BALLOT((cur & LEAF_BIT) == 0u));
        const int stragglers = descend_below < (entering >> ORT_DESCEND_SHIFT) ? descend_below : (entering >> ORT_DESCEND_SHIFT);
        while (!(cur & LEAF_BIT)) {
            ORT_UTIL(sv, 0, true);
            const uint32_t ni = cur & NODE_INDEX_MASK;
            if (WIDE) { /* sv.nodes holds the 4-wide form (DevNode4, 8 float4 each): seven 16-byte loads in flight together */
                const float4 *np = sv.nodes + 8u * ni;
                const float4 lx = np[0], ly = np[1], lz = np[2], hx = np[3], hy = np[4], hz = np[5], cw = np[6];
                visit_node4<COUNTERS, LDS_ENTRIES, BLOCK>(lx, ly, lz, hx, hy, hz, cw, org, inv_d, h.best_t, cur, sp, lds_stack, spill, tid, c);
            } else {
            float4 na, nb, nc, nd;
            if (TREELET && ni < kTreeletNodes) { /* the top of the tree: LDS */
                const float4 *np = tab + kTabTreelet + 4u * ni;
                na = np[0]; nb = np[1]; nc = np[2]; nd = np[3];
            } else {
                const float4 *np = sv.nodes + 4u * ni;
                na = np[0]; nb = np[1]; nc = np[2]; nd = np[3];
            }
            visit_node<COUNTERS, LDS_ENTRIES, BLOCK>(na, nb, nc, nd, org, inv_d, h.best_t, cur, sp, lds_stack, spill, tid, c);
            }
            /* the stragglers of the descend loop would keep the rest of the wave waiting: break out
               and come back for them (their cur / sp carry over) */
            if (ORT_POPC64(ORT_BALLOT(true)) < stragglers) break;
        }
        ORT_PHASE(pr, sv, 5, true);
        if (!(cur & LEAF_BIT)) {
            /* still on an interior node after the early exit above: nothing to do this round */
        } else if (cur == kTraversalDone) {
            tracing = false;
        } else {
            ORT_UTIL(sv, 1, true);
            uint32_t kind = (cur >> 28) & 7u, count = ((cur >> 24) & 15u) + 1u, first = cur & 0x00ffffffu;
            for (uint32_t i = 0; i < count; ++i)
                test_prim<COUNTERS, false>(sv, kind, first + i, org, dir, inv_d, h.best_t, h.hit_n, h.hit_prim, h.phantom_t, h.runner_t, c.tris, c.analytic, excl);
            if (sp == 0) {
                cur = kTraversalDone;
                tracing = false;
            } else {
                sp--;
                /* two real loads behind a branch (the volatile keeps the compiler from merging them into one
                   flat_load of a selected generic pointer): the LDS side becomes a plain ds_read */
                if (sp < LDS_ENTRIES) cur = lds_stack[sp * BLOCK + tid];
                else cur = ((volatile uint32_t *)spill)[sp - LDS_ENTRIES];
            }
            ORT_PHASE(pr, sv, 6, true);
        }
        /* finished: the winner's chain word and material are wanted next (resolve_hit), ask for them now */
        if (ANNOUNCE && !tracing) announce_winner<BLOCK>(sv, h.hit_prim, lds_stack, tid);
        /* when most of the wave has finished its ray, let the finished lanes shade and refill */
        if (refill_below > 0 && ORT_POPC64(ORT_BALLOT(tracing)) < refill_below) break;
    }
    T.cur = cur;
    T.sp = sp;
    return tracing;
}

/* After a traversal: is the winner W of the fast traversal what the reference returns (ray.cpp:788-803)?

   - W's chain admits the ray (chain_verdict): done.
   - A box of W's chain can never be entered (CH_REJECT: missed, or entered below 1e-6 from outside --
     the reference's cylinder boxes do not contain their cylinders): W is invisible to this ray whatever the
     visiting order, so the answer is the best of the OTHER shapes: one more fast traversal with W ignored,
     whose winner is checked the same way.
   - W's leaf box is entered beyond W's own distance with room for another hit in between (CH_UNKNOWN): one
     more fast traversal, limited to that entry distance and ignoring W, settles whether such a hit exists;
     if not, W stands.
   - Anything else (a phantom that could win, a second complication on the same ray): the literal
     breadth-first emulation.
   The extra traversals run here, to completion, for the lanes that need them (1e-5 of the rays of the
   reference's scenes, 1e-2 with slanted cylinders) while the rest of the wave waits: the shape to ignore is
   a local of this rare branch, not a register carried through every ray's traversal. */
template <bool COUNTERS, bool TABS, int LDS_ENTRIES, int BLOCK, bool ANNOUNCED = false, bool WIDE = false>
ORT_D void resolve_hit(const SceneView &sv, const float4 *tab, V3 org, V3 dir, V3 inv_d, uint32_t lane_id, HitState &h, Counters &c, Prof &pr,
                       uint32_t *lds_stack, uint32_t *spill, int tid) {
    /* the winner's chain word and material index: announced by the traversal (in the lane's stack entries 0 and 1),
       or fetched here */
    uint32_t word = 0, mat = 0;
    if (ANNOUNCED) {
        announced_info<BLOCK>(lds_stack, tid, word, mat);
    } else if (h.hit_prim != kNoPrim) {
        const PrimInfo pi = sv.prim_info[info_index(sv, h.hit_prim)];
        word = pi.chain; mat = pi.mat;
    }
    bool stale = false; /* the winner has changed since */
    bool recast = sv.force_fallback_mask != 0xffffffffu && (om_f32_bits(dir.x) & sv.force_fallback_mask) == 0u;
    if (!ORT_RARE(recast)) {
        if (ORT_RARE(h.phantom_t <= h.best_t)) {
            recast = true;
            ORT_STAT(2, 1);
        } else if (h.hit_prim != kNoPrim) {
            float gap = 0.0f;
            const int verdict = chain_verdict(sv, word, org, inv_d, h.best_t, fminf(h.runner_t, h.phantom_t), gap);
            if (ORT_RARE(verdict != CH_ADMIT)) {
                stale = true;
                ORT_STAT(3, verdict == CH_REJECT ? 0 : 1); ORT_STAT(3, 4 + (int)(h.hit_prim >> 28));
                ORT_COUNT(sv.cold->fallback_counters + kDiagFallback + 1, 1ull); /* diagnostics (ORT_DEBUG_FALLBACK): re-traversals */
                /* W waits in the lane's (idle) traversal-stack slots of LDS, not in registers */
                const uint32_t w_prim = h.hit_prim;
                uint32_t *save = lds_stack + tid;
                save[(LDS_ENTRIES - 1) * BLOCK] = om_f32_bits(h.best_t);
                save[(LDS_ENTRIES - 2) * BLOCK] = om_f32_bits(h.hit_n.x);
                save[(LDS_ENTRIES - 3) * BLOCK] = om_f32_bits(h.hit_n.y);
                save[(LDS_ENTRIES - 4) * BLOCK] = om_f32_bits(h.hit_n.z);
                Trav t2;
                t2.cur = 0; t2.sp = 0; t2.inv_d = inv_d;
                /* CH_UNKNOWN: only hits at or before the leaf box's entry matter (the hit tests' "<" must accept t == gap) */
                reset_hit(h, verdict == CH_REJECT ? 3.402823466e+38f : om_bits_f32(om_f32_bits(gap) + 1u));
                prologue_tests<COUNTERS, TABS, true>(sv, tab, org, dir, inv_d, h, c, w_prim);
                (void)traverse<COUNTERS, LDS_ENTRIES - 4, BLOCK, false, false, WIDE>(sv, org, dir, t2, h, lds_stack, spill, tid, 0, 0, c, pr, w_prim);
                if (verdict == CH_UNKNOWN) {
                    if (h.hit_prim != kNoPrim || h.phantom_t <= gap) {
                        recast = true; /* something is there: order decides */
                        ORT_STAT(2, h.hit_prim != kNoPrim ? 2 : 6);
                    } else {           /* nothing there: W stands */
                        h.best_t = om_bits_f32(save[(LDS_ENTRIES - 1) * BLOCK]);
                        h.hit_n = mk(om_bits_f32(save[(LDS_ENTRIES - 2) * BLOCK]), om_bits_f32(save[(LDS_ENTRIES - 3) * BLOCK]),
                                     om_bits_f32(save[(LDS_ENTRIES - 4) * BLOCK]));
                        h.hit_prim = w_prim;
                    }
                } else if (h.phantom_t <= h.best_t) {
                    recast = true;
                    ORT_STAT(2, 3);
                } else if (h.hit_prim != kNoPrim) {
                    float gap2 = 0.0f;
                    if (chain_verdict(sv, sv.prim_info[info_index(sv, h.hit_prim)].chain, org, inv_d, h.best_t, fminf(h.runner_t, h.phantom_t), gap2) != CH_ADMIT) { recast = true; ORT_STAT(2, 4); }
                }
            }
        }
    }
    recast_exactly<COUNTERS>(sv, recast, org, dir, inv_d, lane_id, h, c);
    if (ORT_RARE(stale || recast)) mat = (h.hit_prim != kNoPrim) ? sv.prim_info[info_index(sv, h.hit_prim)].mat : 0u;
    h.hit_mat = (h.hit_prim != kNoPrim) ? mat : 0u;
}

ORT_D void flush_counters(const RenderHot &rv, const Counters &c, bool all) {
    if (all) {
        ORT_COUNT(rv.c->counters + 0, c.paths);
        ORT_COUNT(rv.c->counters + 1, c.rays);
        ORT_COUNT(rv.c->counters + 2, c.nodes);
        ORT_COUNT(rv.c->counters + 3, c.tris);
        ORT_COUNT(rv.c->counters + 4, c.analytic);
    }
}

/* persistent mode: one lane runs jobs until the job space is empty */
template <bool COUNTERS, bool DIFFUSE = false, bool TABS = false, bool IMPLICIT = false, bool WIDE = false>
ORT_D void pt_lane(const SceneView &sv, const RenderHot &rv, const float4 *tab, uint32_t *lds_stack, float *lds_focal, const int tid,
                   const uint32_t lane_id, bool prof_on = false, unsigned long long *pool = nullptr) {
    uint32_t spill[kSpillStack];
    const uint32_t spp_u = IMPLICIT ? ((rv.mode == JOBS_PIXEL) ? rv.c->spp : rv.c->chunk) : 0u; /* samples per (one-pixel) job */
    Prof pr;
    pr.on = prof_on;
#ifndef ORT_HOST_SIM
    if (COUNTERS && prof_on) pr.t = __builtin_amdgcn_s_memtime();
#endif
    PathState P;
    HitState h;
    Trav T;
    Counters c;
    bool tracing = false;
    for (;;) {
        if (!tracing) {
            ORT_UTIL(sv, 3, true);
            ORT_UTIL(sv, 4, P.ps == PS_HIT);
            ORT_PHASE(pr, sv, 7, true);
            if (P.ps == PS_HIT) resolve_hit<COUNTERS, TABS, kLdsStack, kBlock, true, WIDE>(sv, tab, P.org, P.dir, T.inv_d, lane_id, h, c, pr, lds_stack, spill, tid);
            ORT_PHASE(pr, sv, 0, P.ps == PS_HIT);
            tracing = produce_ray<COUNTERS, DIFFUSE, TABS, IMPLICIT>(sv, rv, tab, P, h, c, pr, lds_focal + tid, kBlock, spp_u, nullptr, false, pool);
            if (tracing) {
                begin_ray<COUNTERS, TABS, kLdsStack, kBlock>(sv, tab, P, T, h, c, pr, lds_stack, spill, tid);
                if (COUNTERS) c.rays++;
            }
        }
        if (ORT_BALLOT(P.ps != PS_DONE) == 0ull) break;
        if (tracing) tracing = traverse<COUNTERS, kLdsStack, kBlock, TABS && !WIDE, true, WIDE>(sv, P.org, P.dir, T, h, lds_stack, spill, tid, rv.refill_below, rv.descend_below, c, pr, kNoPrim, tab);
    }
    flush_counters(rv, c, COUNTERS);
}


#ifndef ORT_HOST_SIM
/* ---- ray exchange: whole waves shade, whole waves traverse --------------------------------------------------
 * The plain loop (pt_lane) leaves the traversal loop when fewer than refill_below lanes are still tracing; those
 * stragglers then sit idle through the whole shading pass, and the next traversal loop runs for them and the few
 * new rays that need more than the root.  Here the stragglers are PARKED instead: path, hit and traversal state
 * (kStashVecs float4 = 36 dwords) plus the used part of the LDS stack go to the wave's own L stash in HBM, and the lane takes a parked
 * path whose ray is finished (R stash) or a new job, so that the shading pass runs with all 64 lanes.  When enough
 * rays are parked, the wave parks its finished paths in R, fills ALL lanes from L and traverses -- 64 rays of the
 * expensive kind together, topping up from L as they finish.  Path state travels with the ray, seeds belong to
 * jobs, so which lane or in which order a path is advanced cannot change a bit of the result.
 * Both stashes are private to the wave (wave-uniform tops, ballot-prefix slots): no atomics, no barriers. */
struct Stash {
    float4 *rec;    /* [kStashVecs][cap] */
    uint32_t *stk;  /* [kLdsStack][cap], L only */
    uint32_t cap;
};

ORT_D uint32_t lane_rank(unsigned long long mask) { /* set bits of mask below this lane */
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}

constexpr uint32_t kStashVecs = 9u; /* float4 per parked path */
/* compile-time constants, so that the nine plane offsets of a parked record are literals instead of scalar registers (which were spilled) */
constexpr uint32_t kCapL = 128u, kCapR = 192u; /* parked paths per wave: unfinished rays / finished rays */
static_assert(kLdsStack % 4 == 0, "the L stash carves its stack words out of float4 units (kLdsStack / 4 per parked path)");
ORT_D void stash_store(const Stash &st, uint32_t slot, const PathState &P, const HitState &h, uint32_t cur, int sp, V3 inv_d, const float *focal_cache,
                       uint32_t info_chain = 0u, uint32_t info_mat = 0u) {
    float4 *r = st.rec + slot;
    const uint32_t cap = st.cap;
    r[0] = make_float4(P.org.x, P.org.y, P.org.z, P.dir.x);
    r[cap] = make_float4(P.dir.y, P.dir.z, h.best_t, om_bits_f32(h.hit_prim));
    r[2u * cap] = make_float4(h.hit_n.x, h.hit_n.y, h.hit_n.z, h.phantom_t);
    r[3u * cap] = make_float4(h.runner_t, om_bits_f32(cur), om_bits_f32((uint32_t)sp), om_bits_f32(P.rng));
    r[4u * cap] = make_float4(P.color.x, P.color.y, P.color.z, P.weight.x);
    r[5u * cap] = make_float4(P.weight.y, P.weight.z, P.wo.x, P.wo.y);
    r[6u * cap] = make_float4(P.wo.z, om_bits_f32(P.pxy), om_bits_f32(P.jyp), om_bits_f32(P.sample | (P.primary ? 0x80000000u : 0u)));
    /* 1/d and the pixel's focal point travel too: recomputing them costs more than two more stores and loads */
    r[7u * cap] = make_float4(inv_d.x, inv_d.y, inv_d.z, focal_cache[0]);
    /* a finished ray's announced winner info (R stash; announce_winner) */
    r[8u * cap] = make_float4(focal_cache[kBlock], focal_cache[2 * kBlock], om_bits_f32(info_chain), om_bits_f32(info_mat));
}

ORT_D void stash_load(const Stash &st, uint32_t slot, const RenderHot &rv, PathState &P, HitState &h, Trav &T, float *focal_cache, uint32_t *lds_info) {
    const float4 *r = st.rec + slot;
    const uint32_t cap = st.cap;
    const float4 a = r[0], b = r[cap], c = r[2u * cap], d = r[3u * cap], e = r[4u * cap], f = r[5u * cap], g = r[6u * cap];
    const float4 i = r[7u * cap], j = r[8u * cap];
    P.org = mk(a.x, a.y, a.z); P.dir = mk(a.w, b.x, b.y);
    h.best_t = b.z; h.hit_prim = om_f32_bits(b.w);
    h.hit_n = mk(c.x, c.y, c.z); h.phantom_t = c.w;
    h.runner_t = d.x; T.cur = om_f32_bits(d.y); T.sp = (int)om_f32_bits(d.z); P.rng = om_f32_bits(d.w);
    P.color = mk(e.x, e.y, e.z); P.weight = mk(e.w, f.x, f.y); P.wo = mk(f.z, f.w, g.x);
    P.pxy = om_f32_bits(g.y); P.jyp = om_f32_bits(g.z);
    const uint32_t sm = om_f32_bits(g.w);
    P.sample = sm & 0x7fffffffu; P.primary = (sm >> 31) != 0u;
    P.ps = PS_HIT;
    T.inv_d = mk(i.x, i.y, i.z);
    focal_cache[0] = i.w; focal_cache[kBlock] = j.x; focal_cache[2 * kBlock] = j.y;
    /* where resolve_hit looks for them (an L path refills its stack over them: it has not finished yet) */
    lds_info[0] = om_f32_bits(j.z); lds_info[kBlock] = om_f32_bits(j.w);
}

template <bool COUNTERS, bool DIFFUSE, bool TABS>
ORT_D void pt_lane_x(const SceneView &sv, const RenderHot &rv, const float4 *tab, uint32_t *lds_stack, float *lds_focal, const int tid,
                     const uint32_t lane_id, bool prof_on, unsigned long long *pool) {
    uint32_t spill[kSpillStack];
    Prof pr;
    pr.on = prof_on;
    if (COUNTERS && prof_on) pr.t = __builtin_amdgcn_s_memtime();
    PathState P;
    HitState h;
    Trav T;
    Counters c;
    bool tracing = false;

    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(lane_id >> 6)); /* wave-uniform: the stash addresses stay in scalar registers */
    const uint32_t spp_u = (rv.mode == JOBS_PIXEL) ? rv.c->spp : rv.c->chunk;                /* samples per (one-pixel) job */
    float4 *wbase = rv.c->stash + (size_t)wave * rv.c->stash_wave_f4;
    Stash L, R;
    L.rec = wbase; L.cap = kCapL;
    L.stk = (uint32_t *)(wbase + kStashVecs * kCapL);
    R.rec = wbase + (kStashVecs + (uint32_t)kLdsStack / 4u) * kCapL; R.cap = kCapR; R.stk = nullptr;
    uint32_t ltop = 0, rtop = 0; /* wave-uniform */
    float *focal_cache = lds_focal + tid;
    /* the wave stops parking, and hands its parked paths to free lanes before new jobs, as soon as one of its lanes has
       drawn a job beyond endgame_from: every parked path is a job in progress, and the oldest ones lie at the bottom of the
       LIFO stashes until something drains them -- left to the very end they are a second tail after the job space is empty.
       The lane that draws such a job raises a word of its own in LDS (produce_ray); the wave looks at the 64 words once per
       exchange step.  (Looking at the job counter itself, one lane every 16th step, halved the kernel's speed: 4 096 waves
       reading the one line every job draw of the chip goes through.) */
    uint32_t *late_flag = (uint32_t *)lds_focal + 3 * kBlock + tid;
    *late_flag = 0u;
    bool early_end = false;

    for (;;) {
        if (!early_end) early_end = __ballot(*(volatile uint32_t *)late_flag != 0u) != 0ull;
        /* ---- exchange: every lane is tracing (unfinished ray), done (finished ray, PS_HIT) or free (no path) ---- */
        const unsigned long long m_tr = __ballot(tracing);
        const unsigned long long m_done = __ballot(!tracing && P.ps == PS_HIT);
        const unsigned long long m_free = ~(m_tr | m_done);
        const uint32_t n_tr = (uint32_t)__popcll(m_tr), n_free = (uint32_t)__popcll(m_free);
        /* nothing left but parked rays: traverse them however few */
        const bool drain = m_tr == 0ull && m_done == 0ull && rtop == 0u && __ballot(P.ps != PS_DONE) == 0ull;
        /* once the job space is empty nothing is parked any more (every parked path is a job some lane still has to
           finish): lanes without a path take parked ones, finished rays first, and everything else carries on */
        const bool endgame = early_end || __ballot(P.ps == PS_DONE) != 0ull;
        bool long_phase = !endgame && ltop > 0u && (n_tr + ltop >= rv.c->long_min || drain);
        if (endgame) {
            const bool is_free = !tracing && P.ps != PS_HIT;
            const unsigned long long m_recv = __ballot(is_free);
            const uint32_t rrank = lane_rank(m_recv);
            const bool take_r = is_free && rrank < rtop;
            const bool take_l = is_free && !take_r && rrank - rtop < ltop;
            if (take_r) stash_load(R, rtop - 1u - rrank, rv, P, h, T, focal_cache, lds_stack + tid);
            if (take_l) {
                const uint32_t slot = ltop - 1u - (rrank - rtop);
                stash_load(L, slot, rv, P, h, T, focal_cache, lds_stack + tid);
                for (int lv = 0; lv < T.sp; ++lv) lds_stack[lv * kBlock + tid] = L.stk[(uint32_t)lv * L.cap + slot];
                tracing = true;
            }
            uint32_t n_recv = (uint32_t)__popcll(m_recv);
            const uint32_t from_r = n_recv < rtop ? n_recv : rtop;
            rtop -= from_r;
            n_recv -= from_r;
            ltop -= n_recv < ltop ? n_recv : ltop;
        } else if (long_phase) {
            /* fill the lanes that are not tracing with parked rays; finished paths make room by parking in R */
            uint32_t want = 64u - n_tr;
            if (want > ltop) want = ltop;
            uint32_t need_done = want > n_free ? want - n_free : 0u;
            if (need_done > R.cap - rtop) need_done = R.cap - rtop;
            const bool is_done = !tracing && P.ps == PS_HIT;
            const uint32_t drank = lane_rank(m_done);
            const bool park = is_done && drank < need_done;
            if (park) {
                uint32_t info_chain, info_mat;
                announced_info<kBlock>(lds_stack, tid, info_chain, info_mat);
                stash_store(R, rtop + drank, P, h, 0u, 0, T.inv_d, focal_cache, info_chain, info_mat);
                P.ps = PS_NEED_JOB;
            }
            rtop += need_done < (uint32_t)__popcll(m_done) ? need_done : (uint32_t)__popcll(m_done);
            const bool is_free = !tracing && P.ps != PS_HIT; /* includes the lanes that parked just now */
            const unsigned long long m_recv = __ballot(is_free);
            const uint32_t rrank = lane_rank(m_recv);
            const bool take = is_free && rrank < ltop;
            if (take) {
                const uint32_t slot = ltop - 1u - rrank;
                stash_load(L, slot, rv, P, h, T, focal_cache, lds_stack + tid);
                for (int lv = 0; lv < T.sp; ++lv) lds_stack[lv * kBlock + tid] = L.stk[(uint32_t)lv * L.cap + slot];
                tracing = true;
            }
            const uint32_t n_recv = (uint32_t)__popcll(m_recv);
            ltop -= n_recv < ltop ? n_recv : ltop;
            /* no lane could take a ray (R full, so no finished path could make room) and none is tracing: shade
               instead, which frees lanes */
            if (n_tr == 0u && __ballot(take) == 0ull) long_phase = false;
        } else {
            /* park the stragglers (their stack tail must be in LDS), then hand the free lanes parked finished paths */
            const bool can_park = tracing && T.sp <= kLdsStack && n_tr >= rv.c->park_min;
            const unsigned long long m_park = __ballot(can_park);
            const uint32_t prank = lane_rank(m_park);
            const bool park = can_park && ltop + prank < L.cap;
            if (park) {
                const uint32_t slot = ltop + prank;
                stash_store(L, slot, P, h, T.cur, T.sp, T.inv_d, focal_cache);
                for (int lv = 0; lv < T.sp; ++lv) L.stk[(uint32_t)lv * L.cap + slot] = lds_stack[lv * kBlock + tid];
                tracing = false;
                P.ps = PS_NEED_JOB;
            }
            {
                const uint32_t n_park = (uint32_t)__popcll(m_park), room = L.cap - ltop;
                ltop += n_park < room ? n_park : room;
            }
            const bool is_free = !tracing && P.ps != PS_HIT;
            const unsigned long long m_recv = __ballot(is_free);
            const uint32_t rrank = lane_rank(m_recv);
            const bool take = is_free && rrank < rtop;
            if (take) {
                stash_load(R, rtop - 1u - rrank, rv, P, h, T, focal_cache, lds_stack + tid);
            }
            const uint32_t n_recv = (uint32_t)__popcll(m_recv);
            rtop -= n_recv < rtop ? n_recv : rtop;
        }
        /* ---- shade: only outside a traversal phase, so that it runs with (nearly) all lanes ---- */
        const bool hold = P.ps == PS_NEED_JOB && ltop + rtop >= rv.c->inflight_cap; /* no new job for now */
        if (!long_phase && !tracing && !hold) {
            ORT_UTIL(sv, 3, true);
            ORT_UTIL(sv, 4, P.ps == PS_HIT);
            ORT_PHASE(pr, sv, 7, true);
            if (P.ps == PS_HIT) resolve_hit<COUNTERS, TABS, kLdsStack, kBlock, true>(sv, tab, P.org, P.dir, T.inv_d, lane_id, h, c, pr, lds_stack, spill, tid);
            ORT_PHASE(pr, sv, 0, P.ps == PS_HIT);
            /* near the end of the launch a lane whose job ends draws no new one while the wave still holds parked paths: the
               next exchange step hands it one of those (endgame branch above), so that the stashes are empty when the job
               space is */
            tracing = produce_ray<COUNTERS, DIFFUSE, TABS, true>(sv, rv, tab, P, h, c, pr, focal_cache, kBlock, spp_u, late_flag, early_end && ltop + rtop > 0u, pool);
            if (tracing) {
                begin_ray<COUNTERS, TABS, kLdsStack, kBlock>(sv, tab, P, T, h, c, pr, lds_stack, spill, tid);
                if (COUNTERS) c.rays++;
            }
        }
        if (__ballot(P.ps != PS_DONE || tracing) == 0ull && ltop == 0u && rtop == 0u) break;
        /* in a traversal phase come back for more parked rays when half the lanes have finished; otherwise (and once
           L is empty) when only stragglers are left, which then park */
        const int below = (long_phase && ltop > 0u) ? (int)rv.c->long_refill : rv.refill_below;
        if (tracing) tracing = traverse<COUNTERS, kLdsStack, kBlock, TABS, true>(sv, P.org, P.dir, T, h, lds_stack, spill, tid, below, rv.descend_below, c, pr, kNoPrim, tab);
    }
    flush_counters(rv, c, COUNTERS);
}
#endif /* !ORT_HOST_SIM */

/* ---- wavefront mode -------------------------------------------------------------------------- */
constexpr uint32_t WF_PRIMARY = 8u, WF_HAS_RAY = 16u;
constexpr int kWfLdsStack = 16; /* trace kernel: 16 LDS entries per lane (16 KB per 256-lane block), tail in scratch */
constexpr int kWfSpill = 48;
static_assert(kWfLdsStack - 4 + kWfSpill >= (int)kTreeDepthBudget, "wavefront trace stack must hold a tree of kTreeDepthBudget levels");

/* one slot: resume its path, produce the next ray, store everything back; returns true if a ray was produced */
template <bool COUNTERS>
ORT_D bool wf_shade_slot(const SceneView &sv, const RenderHot &rv, const float4 *tab, const WfView &wf, uint32_t i, Counters &c) {
    uint32_t fl = wf.flags[i];
    PathState P;
    P.ps = (int)(fl & 7u);
    if (P.ps == PS_DONE) return false;
    HitState h;
    if (P.ps != PS_NEED_JOB) {
        float4 a = wf.od0[i];
        float2 b = wf.od1[i];
        float4 q0 = wf.p0[i], q1 = wf.p1[i], q2 = wf.p2[i];
        uint4 q3 = wf.p3[i];
        float4 hh = wf.hit0[i];
        P.org = mk(a.x, a.y, a.z); P.dir = mk(a.w, b.x, b.y);
        P.weight = mk(q0.x, q0.y, q0.z); P.color = mk(q0.w, q1.x, q1.y); P.wo = mk(q1.z, q1.w, q2.x);
        P.rng = om_f32_bits(q2.y); P.sample = om_f32_bits(q2.z); P.spp = om_f32_bits(q2.w);
        P.job_index = q3.x; P.pxy = q3.y; P.jxx = q3.z; P.jyp = q3.w;
        P.primary = (fl & WF_PRIMARY) != 0;
        h.best_t = hh.x; h.hit_n = mk(hh.y, hh.z, hh.w); h.hit_prim = wf.hitp[i];
        h.hit_mat = (P.ps == PS_HIT && h.hit_prim != kNoPrim) ? sv.prim_info[info_index(sv, h.hit_prim)].mat : 0u;
    }
    Prof pr;
    bool tracing = produce_ray<COUNTERS>(sv, rv, tab, P, h, c, pr);
    if (tracing) {
        wf.od0[i] = make_float4(P.org.x, P.org.y, P.org.z, P.dir.x);
        wf.od1[i] = make_float2(P.dir.y, P.dir.z);
        wf.p0[i] = make_float4(P.weight.x, P.weight.y, P.weight.z, P.color.x);
        wf.p1[i] = make_float4(P.color.y, P.color.z, P.wo.x, P.wo.y);
        wf.p2[i] = make_float4(P.wo.z, om_bits_f32(P.rng), om_bits_f32(P.sample), om_bits_f32(P.spp));
        wf.p3[i] = make_uint4(P.job_index, P.pxy, P.jxx, P.jyp);
        if (COUNTERS) c.rays++;
    }
    wf.flags[i] = (uint32_t)P.ps | (P.primary ? WF_PRIMARY : 0u) | (tracing ? WF_HAS_RAY : 0u);
    return tracing;
}

/* one slot: closest hit of its ray, resolved to the reference's answer */
template <bool COUNTERS>
ORT_D void wf_trace_slot(const SceneView &sv, const float4 *tab, const WfView &wf, uint32_t i, uint32_t lane_id, uint32_t *lds_stack,
                         uint32_t *spill, int tid, Counters &c) {
    if (!(wf.flags[i] & WF_HAS_RAY)) return;
    float4 a = wf.od0[i];
    float2 b = wf.od1[i];
    PathState P;
    P.org = mk(a.x, a.y, a.z); P.dir = mk(a.w, b.x, b.y);
    HitState h;
    Trav T;
    Prof pr;
    begin_ray<COUNTERS, false, kWfLdsStack, kBlock>(sv, tab, P, T, h, c, pr, lds_stack, spill, tid);
    traverse<COUNTERS, kWfLdsStack, kBlock>(sv, P.org, P.dir, T, h, lds_stack, spill, tid, 0, 0, c, pr);
    resolve_hit<COUNTERS, false, kWfLdsStack, kBlock>(sv, tab, P.org, P.dir, T.inv_d, lane_id, h, c, pr, lds_stack, spill, tid);
    wf.hit0[i] = make_float4(h.best_t, h.hit_n.x, h.hit_n.y, h.hit_n.z);
    wf.hitp[i] = h.hit_prim;
}

/* pixel = (sum over k of partial[k], in k order) / nchunks for one pixel (CHUNK policy) */
ORT_D void combine_pixel(const RenderHot &rv, unsigned long long idx) {
    uint32_t blk = rv.c->shard_index + (uint32_t)(idx >> 6) * rv.c->shard_count;
    uint32_t pin = (uint32_t)(idx & 63ull);
    int x = (int)((rv.c->block_x0 + blk % rv.c->blocks_w) * 8u + (pin & 7u));
    int y = (int)((rv.c->block_y0 + blk / rv.c->blocks_w) * 8u + (pin >> 3));
    if (x < rv.c->x0 || x >= rv.c->x1 || y < rv.c->y0 || y >= rv.c->y1) return;
    const size_t plane = (size_t)rv.c->my_blocks * 64u * 3u; /* partial planes are packed: idx is the pixel's place in each */
    V3 acc = mk(0, 0, 0);
    for (uint32_t k = 0; k < rv.c->nchunks; ++k) {
        const float *p = rv.c->partial + (size_t)k * plane + 3u * (size_t)idx;
        acc = add(acc, mk(p[0], p[1], p[2]));
    }
    acc = divs(acc, (float)rv.c->nchunks);
    float *o = rv.c->packed_out ? rv.c->out + 3u * (size_t)idx : rv.c->out + 3u * ((size_t)y * (size_t)rv.W + (size_t)x);
    o[0] = acc.x; o[1] = acc.y; o[2] = acc.z;
}

#ifndef ORT_HOST_SIM
#ifndef ORT_WAVES_PER_EU
#define ORT_WAVES_PER_EU 4 /* VGPR budget: 4 waves/SIMD = 128 registers, 7 (diffuse flavour) / 26 (general) spilled; tuned on MI355X: profiles/r01_tuning.md */
#endif
/* DIFFUSE: every surface material of the uploaded scene has Ks = Kt = 0, so the evaluation and pdf
   of the specular / transmission lobes are compiled out (sampling keeps all three lobes: a draw of
   exactly 1.0 still takes the reference's transmission branch).  Same values, fewer registers. */
/* the workgroup's copy of the small read-only tables (SceneView::tab_src -> LDS) */
__device__ __forceinline__ void fill_tab(const SceneView &sv, float4 *lds_tab) {
    for (int i = (int)threadIdx.x; i < kTabF4; i += (int)blockDim.x) lds_tab[i] = sv.tab_src[i];
    __syncthreads();
}

template <bool COUNTERS, bool DIFFUSE, bool TABS, bool IMPLICIT = false, bool WIDE = false>
__global__ void __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(ORT_WAVES_PER_EU, ORT_WAVES_PER_EU)))
pt_persistent(SceneView sv, RenderHot rv) {
    __shared__ uint32_t lds_stack[kLdsStack * kBlock];
    __shared__ float lds_focal[3 * kBlock]; /* focal[component][lane] */
    __shared__ unsigned long long lds_pool[2 * (kBlock / 64)]; /* per wave: the unissued part of its last batch of job indices (draw_job) */
    __shared__ float4 lds_tab[TABS ? kTabF4 : 1];
    if (TABS) fill_tab(sv, lds_tab);
    const bool prof = COUNTERS && sv.util != nullptr && blockIdx.x < 32u;
    if (prof) {
        if (threadIdx.x < 96) g_lds_prof[threadIdx.x] = 0ull;
        if (threadIdx.x < 4) g_lds_prof[96 + threadIdx.x] = __builtin_amdgcn_s_memtime();
        __syncthreads();
    }
    pt_lane<COUNTERS, DIFFUSE, TABS, IMPLICIT, WIDE>(sv, rv, lds_tab, lds_stack, lds_focal, (int)threadIdx.x, blockIdx.x * (uint32_t)kBlock + threadIdx.x, prof,
                                                    wave_job_pool(rv, lds_pool));
    if ((threadIdx.x & 63u) == 0u && rv.c->drain) rv.c->drain[blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6)] = __builtin_amdgcn_s_memrealtime();
    if (prof) {
        __syncthreads();
        if (threadIdx.x < 96 && g_lds_prof[threadIdx.x]) atomicAdd(sv.util + threadIdx.x, g_lds_prof[threadIdx.x]);
    }
}

/* the same with the ray exchange (pt_lane_x): implicit job spaces only, LDS tables required */
template <bool COUNTERS, bool DIFFUSE>
__global__ void __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(ORT_WAVES_PER_EU, ORT_WAVES_PER_EU)))
pt_persistent_x(SceneView sv, RenderHot rv) {
    __shared__ uint32_t lds_stack[kLdsStack * kBlock];
    __shared__ float lds_focal[4 * kBlock]; /* rows 0-2 focal point, row 3 "this lane drew a job near the end of the launch" (u32) */
    __shared__ unsigned long long lds_pool[2 * (kBlock / 64)]; /* per wave: the unissued part of its last batch of job indices (draw_job) */
    __shared__ float4 lds_tab[kTabF4];
    fill_tab(sv, lds_tab);
    const bool prof = COUNTERS && sv.util != nullptr && blockIdx.x < 32u;
    if (prof) {
        if (threadIdx.x < 96) g_lds_prof[threadIdx.x] = 0ull;
        if (threadIdx.x < 4) g_lds_prof[96 + threadIdx.x] = __builtin_amdgcn_s_memtime();
        __syncthreads();
    }
    pt_lane_x<COUNTERS, DIFFUSE, true>(sv, rv, lds_tab, lds_stack, lds_focal, (int)threadIdx.x, blockIdx.x * (uint32_t)kBlock + threadIdx.x, prof,
                                       wave_job_pool(rv, lds_pool));
    if ((threadIdx.x & 63u) == 0u && rv.c->drain) rv.c->drain[blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6)] = __builtin_amdgcn_s_memrealtime();
    if (prof) {
        __syncthreads();
        if (threadIdx.x < 96 && g_lds_prof[threadIdx.x]) atomicAdd(sv.util + threadIdx.x, g_lds_prof[threadIdx.x]);
    }
}

#ifndef ORT_W5_TU /* the five-waves unit only needs the path-trace kernels */
/* wavefront kernels: fixed-size grids, grid-stride over the slots */
template <bool COUNTERS>
__global__ void __launch_bounds__(kBlock) wf_shade(SceneView sv, RenderHot rv, WfView wf, int count_active) {
    const float4 *lds_tab = nullptr; /* the wavefront kernels read the tables from HBM */
    Counters c;
    unsigned long long produced = 0;
    const uint32_t stride = gridDim.x * (uint32_t)kBlock;
    for (uint32_t i = blockIdx.x * (uint32_t)kBlock + threadIdx.x; i < wf.slots; i += stride)
        if (wf_shade_slot<COUNTERS>(sv, rv, lds_tab, wf, i, c)) produced++;
    if (count_active && produced) atomicAdd(wf.active, produced);
    flush_counters(rv, c, COUNTERS);
}

template <bool COUNTERS>
__global__ void __launch_bounds__(kBlock) wf_trace(SceneView sv, RenderHot rv, WfView wf) {
    __shared__ uint32_t lds_stack[kWfLdsStack * kBlock];
    const float4 *lds_tab = nullptr;
    uint32_t spill[kWfSpill];
    Counters c;
    const uint32_t stride = gridDim.x * (uint32_t)kBlock;
    const uint32_t lane_id = blockIdx.x * (uint32_t)kBlock + threadIdx.x;
    for (uint32_t i = lane_id; i < wf.slots; i += stride)
        wf_trace_slot<COUNTERS>(sv, lds_tab, wf, i, lane_id, lds_stack, spill, (int)threadIdx.x, c);
    flush_counters(rv, c, COUNTERS);
}

__global__ void wf_init(WfView wf) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < wf.slots) wf.flags[i] = (uint32_t)PS_NEED_JOB;
    if (i == 0) *wf.active = 0ull;
}

/* per-function evaluation on the device, for the parity tests: records of {u32 op; f32 in[24]}
   -> f32 out[8], op codes as documented in include/ort.h (ort_unit_eval_device) */
__global__ void unit_eval(const uint32_t *records, uint32_t n, float *out) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t *rec = records + 25u * i;
    uint32_t op = rec[0];
    float a[24];
    for (int k = 0; k < 24; ++k) a[k] = om_bits_f32(rec[1 + k]);
    float o[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    auto in3 = [&](int k) { return mk(a[k], a[k + 1], a[k + 2]); };
    V3 n3 = mk(0, 0, 0);
    switch (op) {
    case 1: {
        V3 v0 = in3(0), e1 = sub(in3(3), v0), e2 = sub(in3(6), v0);
        float t = hit_triangle(v0, e1, e2, in3(9), in3(12));
        o[0] = t;
        if (t >= 0.0f) { V3 c = cross(e1, e2); o[1] = c.x; o[2] = c.y; o[3] = c.z; }
    } break;
    case 2: { bool tg; float t = hit_sphere(in3(0), a[3], in3(4), in3(7), n3, tg); o[0] = t; o[1] = n3.x; o[2] = n3.y; o[3] = n3.z; } break;
    case 3: { V3 dd = in3(9); float t = hit_aab(in3(0), in3(3), in3(6), mk(1.0f / dd.x, 1.0f / dd.y, 1.0f / dd.z), n3); o[0] = t; o[1] = n3.x; o[2] = n3.y; o[3] = n3.z; } break;
    case 4: {
        /* host-precomputed frame arrives in a[13..22]: rot rows (9) + |axis| */
        float t = hit_cylinder(in3(0), a[6], in3(13), in3(16), in3(19), a[22], in3(7), in3(10), n3);
        o[0] = t; o[1] = n3.x; o[2] = n3.y; o[3] = n3.z;
    } break;
    case 5: {
        uint32_t seed = om_f32_bits(a[0]);
        Mat m = make_mat(in3(8), in3(11), in3(14), a[17]);
        bool tr;
        V3 wi = sample_brdf(seed, in3(1), in3(4), a[7], m, tr);
        o[0] = wi.x; o[1] = wi.y; o[2] = wi.z; o[3] = tr ? 1.0f : 0.0f; o[4] = om_bits_f32(seed);
    } break;
    case 6: {
        Mat m = make_mat(in3(10), in3(13), in3(16), a[19]);
        o[0] = pdf_brdf(in3(0), in3(3), in3(6), a[9], m);
    } break;
    case 7: {
        Mat m = make_mat(in3(9), in3(12), in3(15), a[18]);
        V3 f = eval_scattering(in3(0), in3(3), in3(6), m, a[19], a[20]);
        o[0] = f.x; o[1] = f.y; o[2] = f.z;
    } break;
    case 8: { V3 r = sample_lobe(in3(0), a[3], a[4]); o[0] = r.x; o[1] = r.y; o[2] = r.z; } break;
    case 9:
        o[0] = ort_sinf(a[0]); o[1] = ort_cosf(a[0]); o[2] = ort_atan2f(a[1], a[0]); o[3] = ort_powf(a[0], a[1]); o[4] = ort_logf(a[0]);
        break;
    case 10: { V3 r = normalize(in3(0)); o[0] = r.x; o[1] = r.y; o[2] = r.z; } break;
    case 11: {
        V3 F = fresnel(in3(0), a[3]);
        o[0] = F.x; o[1] = F.y; o[2] = F.z;
        o[3] = ggx_d(in3(4), in3(7), a[10]);
        o[4] = geom(in3(11), in3(4), in3(7), a[10]);
    } break;
    case 12: { /* RNG: seed -> state after one step, rng_01, rng_between(0, 2pi) from the same seed */
        uint32_t s0 = om_f32_bits(a[0]), s1 = s0, s2 = s0;
        rng_step(s0);
        o[0] = om_bits_f32(s0);
        o[1] = rng_01(s1);
        o[2] = rng_between(s2, 0.0f, 2 * kPi);
        o[3] = om_bits_f32(s2);
        o[4] = om_bits_f32(job_seed(om_f32_bits(a[0]), om_f32_bits(a[1])));
    } break;
    case 13: { /* raw IEEE f32 arithmetic: the bit-exactness premise (div, sqrt, mul, add, sub, u32->f32) */
        o[0] = a[0] / a[1]; o[1] = __builtin_sqrtf(a[0]); o[2] = a[0] * a[1]; o[3] = a[0] + a[1]; o[4] = a[0] - a[1];
        o[5] = (float)om_f32_bits(a[0]); o[6] = a[0] * a[1] + a[2];
    } break;
    default: break;
    }
    for (int k = 0; k < 8; ++k) out[8u * i + k] = o[k];
}

__global__ void combine_chunks(RenderHot rv) {
    unsigned long long idx = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (unsigned long long)rv.c->my_blocks * 64ull) return;
    combine_pixel(rv, idx);
}

#endif /* !ORT_W5_TU */
#endif /* !ORT_HOST_SIM */

} // namespace ORT_NS

#endif /* ORT_LANE_H */
