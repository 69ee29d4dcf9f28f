/*
 * ort_kernels.hip -- host side of the path-trace kernels: scene upload, launch policy, the render call (device_render), and the
 * kernels' four-waves-per-SIMD build (instantiated by the launches below; the lane code itself is ort_lane.h).
 */
#include <algorithm>

#include "ort_lane.h"

#ifndef ORT_HOST_SIM /* tools/host_sim.cpp drives the lane code itself and has no device */
/* the five-waves build of the plain loop (ort_kernels_w5.hip) */
void ort_launch_w5(int diffuse, unsigned int grid, void *stream, const void *sv_bytes, const void *hot_bytes);
size_t ort_w5_sizeof_scene_view();
size_t ort_w5_sizeof_render_hot();

namespace ort {

using namespace ortd;

/* Developer knobs (none changes a result).  The environment is read ONCE, when the scene is uploaded
   (ort_scene_upload); ORT_KNOBS_LIVE=1 -- tests and tuning sweeps that flip a knob between two renders of one uploaded
   scene -- reads it again at every render.  -1 = not set: the launch policy decides. */
struct Knobs {
    uint32_t force_fallback_mask = 0xffffffffu; /* ORT_DEBUG_FORCE_FALLBACK */
    bool debug_util = false, debug_fallback = false, debug_drain = false; /* ORT_DEBUG_UTIL, ORT_DEBUG_FALLBACK, ORT_DEBUG_DRAIN */
    int cache_resident = -1;   /* ORT_CACHE_RESIDENT */
    int refill_below = -1, descend_below = -1; /* ORT_REFILL_BELOW, ORT_DESCEND_BELOW */
    bool wavefront = false;    /* ORT_MODE=wavefront */
    bool general_kernel = false; /* ORT_KERNEL=general */
    int lds_tables = -1;       /* ORT_LDS_TABLES */
    int exchange = -1;         /* ORT_EXCHANGE */
    int long_min = -1, long_refill = -1, inflight_cap = -1, park_min = -1; /* ORT_LONG_MIN, ORT_LONG_REFILL, ORT_INFLIGHT_CAP, ORT_PARK_MIN */
    int lpt = -1;              /* ORT_LPT=0: CHUNK jobs issued chunk-major (rounds 1-2) instead of block-major */
    int wide = -1;             /* ORT_WIDE=1: traverse the 4-wide form of the tree (default: never) */
    int waves5 = -1;           /* ORT_WAVES5=0 / 1: the plain loop's five-waves-per-SIMD build (default: all-lobes flavour, trees that leave the L2) */
    int endgame_jobs = -1;     /* ORT_ENDGAME_JOBS: the ray exchange drains its stashes over the last n/4 jobs per lane (default 16 = four jobs) */
    int blocks_per_cu = -1;    /* ORT_BLOCKS_PER_CU (takes effect at upload) */
    int job_batch = -1, batch_tail = -1; /* ORT_JOB_BATCH: job indices a wave draws at a time (0: one draw per job); ORT_BATCH_TAIL: ... until this many jobs per lane are left */
};
static int env_int(const char *name, int unset = -1) {
    const char *e = getenv(name);
    return e ? atoi(e) : unset;
}
static Knobs read_knobs() {
    Knobs k;
    const char *e;
    if ((e = getenv("ORT_DEBUG_FORCE_FALLBACK"))) k.force_fallback_mask = (uint32_t)strtoul(e, nullptr, 0);
    k.debug_util = getenv("ORT_DEBUG_UTIL") != nullptr;
    k.debug_fallback = getenv("ORT_DEBUG_FALLBACK") != nullptr;
    k.debug_drain = getenv("ORT_DEBUG_DRAIN") != nullptr;
    k.cache_resident = env_int("ORT_CACHE_RESIDENT");
    k.refill_below = env_int("ORT_REFILL_BELOW");
    k.descend_below = env_int("ORT_DESCEND_BELOW");
    k.wavefront = (e = getenv("ORT_MODE")) && strcmp(e, "wavefront") == 0;
    k.general_kernel = (e = getenv("ORT_KERNEL")) && strcmp(e, "general") == 0;
    k.lds_tables = env_int("ORT_LDS_TABLES");
    k.exchange = env_int("ORT_EXCHANGE");
    k.long_min = env_int("ORT_LONG_MIN");
    k.long_refill = env_int("ORT_LONG_REFILL");
    k.inflight_cap = env_int("ORT_INFLIGHT_CAP");
    k.park_min = env_int("ORT_PARK_MIN");
    k.lpt = env_int("ORT_LPT");
    k.wide = env_int("ORT_WIDE");
    k.waves5 = env_int("ORT_WAVES5");
    k.endgame_jobs = env_int("ORT_ENDGAME_JOBS");
    k.blocks_per_cu = env_int("ORT_BLOCKS_PER_CU");
    k.job_batch = env_int("ORT_JOB_BATCH");
    k.batch_tail = env_int("ORT_BATCH_TAIL");
    return k;
}

struct DeviceScene {
    int device = -1;
    Knobs knobs;
    void *nodes = nullptr, *tris = nullptr, *spheres = nullptr, *boxes = nullptr, *cyls = nullptr, *materials = nullptr;
    void *nodes4 = nullptr; /* the 4-wide form of the tree (uploaded when it exists and its depth fits the traversal stacks) */
    void *prim_info = nullptr;
    uint32_t info_box = 0, info_cyl = 0, info_sphere = 0;
    void *light_is_sphere = nullptr;
    void *tab = nullptr; /* image of the LDS tables (kTabF4 float4) */
    void *cold = nullptr; /* SceneCold */
    void *rv_dev = nullptr; /* the RenderView of the render in flight (RenderHot::c) */
    uint32_t tab_flags = 0;
    uint32_t light_count = 0;
    bool diffuse_only = false; /* no surface material can enter the specular / transmission blocks */
    void *ref_nodes = nullptr, *ref_recs = nullptr, *chain_boxes = nullptr;
    void *tri_order = nullptr, *sphere_order = nullptr, *box_order = nullptr, *cyl_order = nullptr;
    void *bfs_pool = nullptr, *bfs_locks = nullptr;
    uint32_t bfs_queue_cap = 0, bfs_queue_count = 0;
    unsigned int max_blocks = 0;
    unsigned long long *ctrl = nullptr; /* [0] next_job, [1..5] counters */
    float *partial = nullptr;
    size_t partial_bytes = 0;
    float *staging = nullptr;
    size_t staging_bytes = 0;
    void *jobs = nullptr;
    size_t jobs_bytes = 0;
    void *states = nullptr;
    size_t states_bytes = 0;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    hipEvent_t ev_done = nullptr; /* end of the last render enqueued on this scene */
    bool inflight = false;        /* that render was returned from without waiting (device form, stats == NULL) */
    int cu_count = 0;
    void *stash = nullptr; /* ray exchange: the waves' stashes */
    size_t stash_bytes = 0;
    void *drain = nullptr; /* ORT_DEBUG_DRAIN: per-wave end times */
    size_t drain_bytes = 0;
    void *wf_mem = nullptr; /* wavefront state, carved into the WfView arrays */
    size_t wf_bytes = 0;
    unsigned long long *h_active = nullptr; /* pinned */
};

/* ---- host side --------------------------------------------------------------------------- */
#define ORT_HIP(call)                                                                          \
    do {                                                                                       \
        hipError_t e_ = (call);                                                                \
        if (e_ != hipSuccess) {                                                                \
            *err = std::string(#call) + ": " + hipGetErrorString(e_);                          \
            return ORT_ERR_HIP;                                                                \
        }                                                                                      \
    } while (0)

int device_count(int *n, std::string *err) {
    int c = 0;
    hipError_t e = hipGetDeviceCount(&c);
    if (e != hipSuccess) { *n = 0; *err = std::string("hipGetDeviceCount: ") + hipGetErrorString(e); return ORT_ERR_NO_DEVICE; }
    *n = c;
    return ORT_OK;
}

template <typename T>
static int upload_vec(const std::vector<T> &v, void **dst, std::string *err) {
    size_t bytes = v.size() * sizeof(T);
    ORT_HIP(hipMalloc(dst, bytes ? bytes : 16));
    if (bytes) ORT_HIP(hipMemcpy(*dst, v.data(), bytes, hipMemcpyHostToDevice));
    return ORT_OK;
}

void device_release(Scene *scene) {
    DeviceScene *d = scene->dev;
    if (!d) return;
    (void)hipSetDevice(d->device);
    void *ptrs[] = {d->nodes, d->nodes4, d->tris, d->spheres, d->boxes, d->cyls, d->prim_info,
                    d->materials, d->light_is_sphere, d->tab, d->cold, d->rv_dev, d->ref_nodes, d->ref_recs, d->chain_boxes, d->tri_order, d->sphere_order, d->box_order, d->cyl_order, d->bfs_pool, d->bfs_locks, d->ctrl, d->partial, d->staging, d->jobs, d->states};
    for (void *p : ptrs)
        if (p) (void)hipFree(p);
    if (d->wf_mem) (void)hipFree(d->wf_mem);
    if (d->stash) (void)hipFree(d->stash);
    if (d->drain) (void)hipFree(d->drain);
    if (d->h_active) (void)hipHostFree(d->h_active);
    if (d->ev0) (void)hipEventDestroy(d->ev0);
    if (d->ev1) (void)hipEventDestroy(d->ev1);
    if (d->ev_done) (void)hipEventDestroy(d->ev_done);
    delete d;
    scene->dev = nullptr;
}

int device_upload(Scene *scene, int device, std::string *err) {
    int n = 0;
    int rc = device_count(&n, err);
    if (rc != ORT_OK) return rc;
    if (n <= 0) { *err = "no HIP device visible"; return ORT_ERR_NO_DEVICE; }
    if (device < 0 || device >= n) { *err = "device index out of range"; return ORT_ERR_INVALID; }
    device_release(scene);
    ORT_HIP(hipSetDevice(device));
    DeviceScene *d = new DeviceScene();
    d->device = device;
    scene->dev = d;
    const Tree &t = scene->tree;
    if ((rc = upload_vec(t.nodes, &d->nodes, err))) return rc;
    /* a traversal of the wide tree stacks at most three entries per level; the smallest stack is resolve_hit's re-traversal */
    if (!t.nodes4.empty() && 3u * t.max_depth4 <= (uint32_t)(kLdsStack - 4 + kSpillStack))
        if ((rc = upload_vec(t.nodes4, &d->nodes4, err))) return rc;
    if ((rc = upload_vec(t.tris, &d->tris, err))) return rc;
    if ((rc = upload_vec(t.spheres, &d->spheres, err))) return rc;
    if ((rc = upload_vec(t.boxes, &d->boxes, err))) return rc;
    if ((rc = upload_vec(t.cyls, &d->cyls, err))) return rc;
    {
        std::vector<PrimInfo> info;
        build_prim_info(t, scene->ref, info, d->info_box, d->info_cyl, d->info_sphere);
        if ((rc = upload_vec(info, &d->prim_info, err))) return rc;
    }
    std::vector<DevMaterial> mats(scene->materials.size());
    for (size_t i = 0; i < mats.size(); ++i) mats[i] = make_dev_material(scene->materials[i]);
    if ((rc = upload_vec(mats, &d->materials, err))) return rc;
    d->diffuse_only = true; /* the four guards of eval_scattering / pdf_brdf, for every material a path can scatter on */
    for (size_t i = 1; i < mats.size(); ++i) {
        const DevMaterial &dm = mats[i];
        if (dm.is_light) continue;
        const float ks2 = dm.specular[0] * dm.specular[0] + dm.specular[1] * dm.specular[1] + dm.specular[2] * dm.specular[2];
        const float kt2 = dm.transmission[0] * dm.transmission[0] + dm.transmission[1] * dm.transmission[1] + dm.transmission[2] * dm.transmission[2];
        if (ks2 > 0.0f || kt2 > 0.0f || dm.ps_c > 0.0f || dm.pt_c > 0.0f) d->diffuse_only = false;
    }
    std::vector<uint32_t> lis(scene->lights.size());
    for (size_t i = 0; i < lis.size(); ++i) lis[i] = (scene->lights[i].type == 1u) ? 1u : 0u;
    d->light_count = (uint32_t)lis.size();
    if ((rc = upload_vec(lis, &d->light_is_sphere, err))) return rc;
    {
        /* the image of the LDS tables: root node, prologue shapes, light flags, materials */
        std::vector<float4> tab((size_t)kTabF4, make_float4(0, 0, 0, 0));
        uint32_t *tw = (uint32_t *)tab.data();
        d->tab_flags = 0;
        if (!t.nodes.empty()) memcpy(&tab[kTabRoot], &t.nodes[0], sizeof(DevNode));
        {
            /* the top of the fast tree; slots beyond the tree's size are never addressed */
            const size_t nt = t.nodes.size() < (size_t)kTreeletNodes ? t.nodes.size() : (size_t)kTreeletNodes;
            if (nt) memcpy(&tab[kTabTreelet], t.nodes.data(), nt * sizeof(DevNode));
        }
        if (2u * t.pro_boxes + t.pro_spheres + 4u * t.pro_cyls <= (uint32_t)kTabProCap) {
            float4 *q = &tab[kTabPro];
            if (t.pro_boxes) memcpy(q, t.boxes.data(), (size_t)t.pro_boxes * sizeof(DevBox));
            q += 2u * t.pro_boxes;
            if (t.pro_spheres) memcpy(q, t.spheres.data(), (size_t)t.pro_spheres * sizeof(DevSphere));
            q += t.pro_spheres;
            if (t.pro_cyls) memcpy(q, t.cyls.data(), (size_t)t.pro_cyls * sizeof(DevCyl));
            d->tab_flags |= TAB_PRO;
        }
        if (lis.size() <= (size_t)kTabLightCap) {
            if (!lis.empty()) memcpy(tw + 4 * kTabLights, lis.data(), lis.size() * 4u);
            d->tab_flags |= TAB_LIGHTS;
        }
        if (mats.size() <= (size_t)kTabMatCap) {
            memcpy(&tab[kTabMats], mats.data(), mats.size() * sizeof(DevMaterial));
            d->tab_flags |= TAB_MATS;
        }
        if ((rc = upload_vec(tab, &d->tab, err))) return rc;
    }
    const RefTree &rt = scene->ref;
    if ((rc = upload_vec(rt.nodes, &d->ref_nodes, err))) return rc;
    if ((rc = upload_vec(rt.recs, &d->ref_recs, err))) return rc;
    if ((rc = upload_vec(rt.chain_boxes, &d->chain_boxes, err))) return rc;
    if ((rc = upload_vec(rt.tri_order, &d->tri_order, err))) return rc;
    if ((rc = upload_vec(rt.sphere_order, &d->sphere_order, err))) return rc;
    if ((rc = upload_vec(rt.box_order, &d->box_order, err))) return rc;
    if ((rc = upload_vec(rt.cyl_order, &d->cyl_order, err))) return rc;
    ORT_HIP(hipMalloc((void **)&d->ctrl, 128 * sizeof(unsigned long long)));
    ORT_HIP(hipMemset(d->ctrl, 0, 128 * sizeof(unsigned long long)));
    ORT_HIP(hipEventCreate(&d->ev0));
    ORT_HIP(hipEventCreate(&d->ev1));
    ORT_HIP(hipEventCreateWithFlags(&d->ev_done, hipEventDisableTiming));
    hipDeviceProp_t prop;
    ORT_HIP(hipGetDeviceProperties(&prop, device));
    d->cu_count = prop.multiProcessorCount;
    /* persistent grid: 4 workgroups of 256 lanes per CU */
    {
        d->knobs = read_knobs();
        unsigned int per_cu = d->knobs.blocks_per_cu > 0 ? (unsigned int)d->knobs.blocks_per_cu : 4u; /* resident workgroups per CU (4 = one wave per SIMD each) */
        if (per_cu < 1u || per_cu > 8u) per_cu = 4u;
        d->max_blocks = (unsigned int)(d->cu_count > 0 ? d->cu_count : 256) * per_cu;
    }
    /* fallback queues: one entry per reference-tree node each, as many as fit the budget */
    d->bfs_queue_cap = (uint32_t)rt.nodes.size() + 8u;
    size_t fit = kBfsPoolBytes / ((size_t)d->bfs_queue_cap * sizeof(uint32_t));
    d->bfs_queue_count = (uint32_t)(fit < 16 ? 16 : (fit > kBfsPoolQueues ? kBfsPoolQueues : fit));
    ORT_HIP(hipMalloc(&d->bfs_pool, (size_t)d->bfs_queue_count * d->bfs_queue_cap * sizeof(uint32_t)));
    ORT_HIP(hipMalloc(&d->bfs_locks, (size_t)d->bfs_queue_count * kBfsLockStride * sizeof(uint32_t)));
    ORT_HIP(hipMemset(d->bfs_locks, 0, (size_t)d->bfs_queue_count * kBfsLockStride * sizeof(uint32_t)));
    ORT_HIP(hipHostMalloc((void **)&d->h_active, sizeof(unsigned long long)));
    {
        SceneCold cold{};
        cold.ref_nodes = (const float4 *)d->ref_nodes; cold.ref_recs = (const uint32_t *)d->ref_recs;
        cold.tri_order = (const uint32_t *)d->tri_order; cold.sphere_order = (const uint32_t *)d->sphere_order;
        cold.box_order = (const uint32_t *)d->box_order; cold.cyl_order = (const uint32_t *)d->cyl_order;
        cold.bfs_pool = (uint32_t *)d->bfs_pool; cold.bfs_locks = (uint32_t *)d->bfs_locks;
        cold.bfs_queue_cap = d->bfs_queue_cap; cold.bfs_queue_count = d->bfs_queue_count;
        cold.fallback_counters = d->ctrl + 6;
        ORT_HIP(hipMalloc(&d->rv_dev, sizeof(RenderView)));
        ORT_HIP(hipMalloc(&d->cold, sizeof(SceneCold)));
        ORT_HIP(hipMemcpy(d->cold, &cold, sizeof(SceneCold), hipMemcpyHostToDevice));
    }
    return ORT_OK;
}

int device_unit_eval(int device, const void *records, uint32_t n, float *out, std::string *err) {
    int count = 0;
    int rc = device_count(&count, err);
    if (rc != ORT_OK) return rc;
    if (device < 0 || device >= count) { *err = "no such HIP device"; return ORT_ERR_NO_DEVICE; }
    ORT_HIP(hipSetDevice(device));
    void *d_rec = nullptr, *d_out = nullptr;
    ORT_HIP(hipMalloc(&d_rec, (size_t)n * 100u + 16));
    ORT_HIP(hipMalloc(&d_out, (size_t)n * 32u + 16));
    ORT_HIP(hipMemcpy(d_rec, records, (size_t)n * 100u, hipMemcpyHostToDevice));
    if (n) hipLaunchKernelGGL(unit_eval, dim3((n + 63) / 64), dim3(64), 0, 0, (const uint32_t *)d_rec, n, (float *)d_out);
    ORT_HIP(hipGetLastError());
    ORT_HIP(hipMemcpy(out, d_out, (size_t)n * 32u, hipMemcpyDeviceToHost));
    ORT_HIP(hipFree(d_rec));
    ORT_HIP(hipFree(d_out));
    return ORT_OK;
}

/* the 8x8 blocks a PIXEL / CHUNK render enumerates: the whole grid in its global numbering when sharded (that is
   what block_id % world == rank refers to), only the blocks under the rect on one GPU */
struct BlockGrid { uint32_t blocks_w, block_x0, block_y0, my_blocks, shard_index, shard_count; };
static BlockGrid block_grid_for(const ort_render_params *p) {
    BlockGrid g;
    g.shard_count = p->shard_count > 1 ? p->shard_count : 1;
    g.shard_index = p->shard_count > 1 ? p->shard_index : 0;
    g.blocks_w = (uint32_t)((p->width + 7) / 8);
    uint32_t total = g.blocks_w * (uint32_t)((p->height + 7) / 8);
    g.block_x0 = g.block_y0 = 0;
    if (g.shard_count == 1 && p->x1 > p->x0 && p->y1 > p->y0) {
        g.block_x0 = (uint32_t)(p->x0 / 8);
        g.block_y0 = (uint32_t)(p->y0 / 8);
        g.blocks_w = (uint32_t)((p->x1 + 7) / 8) - g.block_x0;
        total = g.blocks_w * ((uint32_t)((p->y1 + 7) / 8) - g.block_y0);
    }
    g.my_blocks = (total > g.shard_index) ? (total - g.shard_index + g.shard_count - 1) / g.shard_count : 0;
    return g;
}

uint64_t shard_block_count(const ort_render_params *p) { return block_grid_for(p).my_blocks; }

uint64_t render_workspace_bytes(const ort_render_params *p) {
    if (p->policy != ORT_POLICY_CHUNK || p->chunk == 0) return 0;
    uint64_t nch = p->spp / p->chunk;
    return nch * (uint64_t)block_grid_for(p).my_blocks * 64ull * 12ull; /* partial planes hold this shard's blocks only */
}

static int ensure(void **ptr, size_t *have, size_t need, std::string *err) {
    if (*have >= need && *ptr) return ORT_OK;
    if (*ptr) ORT_HIP(hipFree(*ptr));
    *ptr = nullptr;
    *have = 0;
    ORT_HIP(hipMalloc(ptr, need ? need : 16));
    *have = need;
    return ORT_OK;
}

/* wavefront mode: all slots alternate between wf_shade (produce the next ray) and wf_trace (closest
   hit) until no slot produces a ray any more.  The host only learns "finished" by reading a
   counter, so it launches iterations in batches and checks after each batch. */
template <bool COUNTERS>
static int launch_wavefront(DeviceScene *d, const SceneView &sv, const RenderView &rvf, const RenderHot &rv, hipStream_t stream, std::string *err) {
    const uint32_t kMaxSlots = 1u << 21;
    unsigned long long want = rvf.job_count < kMaxSlots ? rvf.job_count : kMaxSlots;
    uint32_t S = (uint32_t)((want + kBlock - 1) / kBlock) * kBlock;
    if (S == 0) S = kBlock;
    const size_t per_slot = 16 + 8 + 16 + 4 + 16 + 16 + 16 + 16 + 4;
    size_t need = (size_t)S * per_slot + 4096;
    int rc = ensure(&d->wf_mem, &d->wf_bytes, need, err);
    if (rc) return rc;
    WfView wf{};
    wf.slots = S;
    char *base = (char *)d->wf_mem;
    auto carve = [&](size_t bytes) { char *p = base; base += (bytes + 255) & ~(size_t)255; return p; };
    wf.active = (unsigned long long *)carve(256);
    wf.od0 = (float4 *)carve((size_t)S * 16); wf.hit0 = (float4 *)carve((size_t)S * 16);
    wf.p0 = (float4 *)carve((size_t)S * 16); wf.p1 = (float4 *)carve((size_t)S * 16); wf.p2 = (float4 *)carve((size_t)S * 16);
    wf.p3 = (uint4 *)carve((size_t)S * 16);
    wf.od1 = (float2 *)carve((size_t)S * 8);
    wf.hitp = (uint32_t *)carve((size_t)S * 4); wf.flags = (uint32_t *)carve((size_t)S * 4);
    if ((size_t)(base - (char *)d->wf_mem) > d->wf_bytes) { *err = "internal: wavefront state carve overflow"; return ORT_ERR_INVALID; }

    unsigned int blocks = S / kBlock;
    unsigned int grid = blocks < d->max_blocks * 2u ? blocks : d->max_blocks * 2u;
    hipLaunchKernelGGL(wf_init, dim3(blocks), dim3(kBlock), 0, stream, wf);
    ORT_HIP(hipGetLastError());
    const int batch = 32;
    for (;;) {
        for (int it = 0; it < batch; ++it) {
            int last = (it == batch - 1);
            if (last) ORT_HIP(hipMemsetAsync(wf.active, 0, sizeof(unsigned long long), stream));
            hipLaunchKernelGGL(wf_shade<COUNTERS>, dim3(grid), dim3(kBlock), 0, stream, sv, rv, wf, last);
            hipLaunchKernelGGL(wf_trace<COUNTERS>, dim3(grid), dim3(kBlock), 0, stream, sv, rv, wf);
        }
        ORT_HIP(hipGetLastError());
        ORT_HIP(hipMemcpyAsync(d->h_active, wf.active, sizeof(unsigned long long), hipMemcpyDeviceToHost, stream));
        ORT_HIP(hipStreamSynchronize(stream));
        if (*d->h_active == 0ull) break;
    }
    return ORT_OK;
}

int device_render(Scene *scene, const ort_render_params *p, const ort_tile_job *jobs, uint32_t job_count, void *d_out,
                  float *h_out, void *stream_v, uint32_t *final_states, ort_stats *stats, std::string *err) {
    DeviceScene *d = scene->dev;
    if (!d) { *err = "scene is not uploaded to a device (ort_scene_upload)"; return ORT_ERR_NO_DEVICE; }
    ORT_HIP(hipSetDevice(d->device));
    hipStream_t stream = (hipStream_t)stream_v;
    const bool packed_out = (p->flags & ORT_RENDER_PACKED) != 0 && !jobs;
    const size_t image_bytes = packed_out ? (size_t)block_grid_for(p).my_blocks * 768u : (size_t)p->width * (size_t)p->height * 12u;
    int rc;
    /* One render at a time per scene: the job counter, the work counters, the partial planes and the stashes belong
       to the scene.  A render that was returned from without waiting is waited for here, and its tripwire checked. */
    if (d->inflight) {
        ORT_HIP(hipEventSynchronize(d->ev_done));
        d->inflight = false;
        unsigned long long ovf = 0;
        ORT_HIP(hipMemcpy(&ovf, d->ctrl + 7, sizeof(ovf), hipMemcpyDeviceToHost));
        if (ovf) { *err = "the previous render on this scene overflowed a reference-order fallback queue"; return ORT_ERR_UNSUPPORTED; }
    }

    float *out = (float *)d_out;
    if (!out) {
        if ((rc = ensure((void **)&d->staging, &d->staging_bytes, image_bytes, err))) return rc;
        out = d->staging;
        if (h_out) ORT_HIP(hipMemcpyAsync(out, h_out, image_bytes, hipMemcpyHostToDevice, stream));
    }

    SceneView sv{};
    sv.nodes = (const float4 *)d->nodes; sv.tris = (const float4 *)d->tris;
    sv.spheres = (const float4 *)d->spheres; sv.boxes = (const float4 *)d->boxes; sv.cyls = (const float4 *)d->cyls;
    sv.prim_info = (const PrimInfo *)d->prim_info;
    sv.info_box = d->info_box; sv.info_cyl = d->info_cyl; sv.info_sphere = d->info_sphere;
    sv.materials = (const float4 *)d->materials;
    sv.light_is_sphere = (const uint32_t *)d->light_is_sphere;
    sv.tab_src = (const float4 *)d->tab;
    sv.tab_flags = d->tab_flags;
    sv.light_count = d->light_count;
    sv.pro_boxes = scene->tree.pro_boxes;
    sv.pro_spheres = scene->tree.pro_spheres;
    sv.pro_cyls = scene->tree.pro_cyls;
    sv.chain_boxes = (const float4 *)d->chain_boxes;
    if (getenv("ORT_KNOBS_LIVE")) { const int keep = d->knobs.blocks_per_cu; d->knobs = read_knobs(); d->knobs.blocks_per_cu = keep; }
    const Knobs &kn = d->knobs;
    sv.force_fallback_mask = kn.force_fallback_mask;
    sv.cold = (const ORT_CONSTANT_AS SceneCold *)d->cold;
    const bool want_util = kn.debug_util; /* developer diagnostics, counters build only */
    sv.util = want_util ? d->ctrl + 8 : nullptr;
    ort_camera cam;
    camera_basis(*scene, p->width, p->height, &cam);
    memcpy(sv.cam, &cam, sizeof(cam));

    RenderView rv{};
    rv.W = p->width; rv.H = p->height;
    rv.x0 = p->x0; rv.y0 = p->y0; rv.x1 = p->x1; rv.y1 = p->y1;
    rv.seed = p->seed; rv.spp = p->spp; rv.chunk = p->chunk; rv.rr = p->rr;
    rv.out = out;
    bool cache_resident_tree = true;
    {
        /* tuning knobs; results do not depend on them.  Defaults tuned on MI355X (profiles/r01_tuning.md)
           separately for trees that stay in L2 and trees that do not */
        const size_t fast_tree_bytes = scene->tree.nodes.size() * sizeof(DevNode) + scene->tree.tris.size() * sizeof(DevTri);
        /* ORT_CACHE_RESIDENT, A/B runs: treat the tree as (not) cache-resident */
        const bool cache_resident = kn.cache_resident >= 0 ? kn.cache_resident != 0 : fast_tree_bytes <= (size_t)(16u << 20);
        cache_resident_tree = cache_resident;
        rv.refill_below = kn.refill_below >= 0 ? kn.refill_below : (cache_resident ? 16 : 32); /* 12 until the block-major issue (round 3: 8-way shard 64.4 -> 63.8 ms) */
        if (rv.refill_below < 1) rv.refill_below = 1;
        if (rv.refill_below > 64) rv.refill_below = 64;
        /* cache-resident trees (bunny room: 6 MB): 8, worth +10 %.  Trees that leave the 8 x 4 MB of L2 (the 1M-triangle
           scene, 86 MB): 16 and a later refill (32): the waits are longer there, so leaving the loops costs more
           (3840x2160, 256 spp: 1 272 Mpaths/s; with the small-tree values 1 100; profiles/r02_tuning.md) */
        rv.descend_below = kn.descend_below >= 0 ? kn.descend_below : (cache_resident ? 8 : 16);
        if (rv.descend_below < 0) rv.descend_below = 0;
        if (rv.descend_below > 64) rv.descend_below = 64;
    }
    rv.next_job = d->ctrl;
    rv.counters = d->ctrl + 1;
    {
        const BlockGrid g = block_grid_for(p);
        rv.shard_count = g.shard_count; rv.shard_index = g.shard_index;
        rv.blocks_w = g.blocks_w; rv.block_x0 = g.block_x0; rv.block_y0 = g.block_y0;
        rv.my_blocks = g.my_blocks;
    }
    rv.packed_out = (p->flags & ORT_RENDER_PACKED) != 0 && !jobs;

    if (jobs) {
        rv.mode = JOBS_EXPLICIT;
        rv.job_count = job_count;
        if ((rc = ensure(&d->jobs, &d->jobs_bytes, (size_t)job_count * sizeof(ort_tile_job), err))) return rc;
        /* synchronous: the caller's job list may be gone when this call returns */
        ORT_HIP(hipStreamSynchronize(stream));
        ORT_HIP(hipMemcpy(d->jobs, jobs, (size_t)job_count * sizeof(ort_tile_job), hipMemcpyHostToDevice));
        rv.jobs = (const ort_tile_job *)d->jobs;
        if (final_states) {
            if ((rc = ensure(&d->states, &d->states_bytes, (size_t)job_count * 4u, err))) return rc;
            rv.final_states = (uint32_t *)d->states;
        }
    } else if (p->policy == ORT_POLICY_PIXEL) {
        rv.mode = JOBS_PIXEL;
        rv.nchunks = 1;
        rv.job_count = (unsigned long long)rv.my_blocks * 64ull;
    } else {
        rv.mode = JOBS_CHUNK;
        rv.nchunks = p->spp / p->chunk;
        rv.job_count = (unsigned long long)rv.my_blocks * 64ull * rv.nchunks;
        size_t need = (size_t)rv.nchunks * (size_t)rv.my_blocks * 64u * 12u; /* = render_workspace_bytes(p) */
        if ((rc = ensure((void **)&d->partial, &d->partial_bytes, need, err))) return rc;
        rv.partial = d->partial;
    }

    const bool counters = (p->flags & ORT_RENDER_COUNTERS) != 0;
    ORT_HIP(hipMemsetAsync(d->ctrl, 0, 128 * sizeof(unsigned long long), stream));
    const bool wavefront = kn.wavefront; /* ORT_MODE=wavefront; results are identical */
    const bool diffuse = d->diffuse_only && !kn.general_kernel; /* ORT_KERNEL=general forces the all-lobes kernel (A/B runs; same results) */
    /* TABS: the scene's small tables all fit their LDS slots (every scene of this repository); otherwise HBM */
    const uint32_t all_tabs = TAB_PRO | TAB_LIGHTS | TAB_MATS;
    const bool tabs = (d->tab_flags & all_tabs) == all_tabs && kn.lds_tables != 0; /* ORT_LDS_TABLES=0: read them from HBM anyway (A/B runs; same results) */
    /* the plain loop of implicit job spaces exists at FIVE waves per SIMD as well (ort_kernels_w5.hip: 96 registers, 20 LDS stack
       entries, machine LICM off).  Same call, four / five waves: analytic scene 3 302 / 3 514 Mpaths/s, glass room 3 625 / 3 848,
       testscene 2 790 / 2 885, 1M-triangle scene 1 401 / 1 500 -- the all-lobes flavour and trees that leave the L2 take it.  The
       diffuse flavour on a cache-resident tree does not: bunny room whole frame 4 729 / 4 712, its 4- / 8-way shards 117.0 / 119.2 and
       63.5 / 65.9 ms (a quarter more lanes, a quarter fewer jobs per lane: the tail weighs more); nor the ray exchange (5 053 / 4 949).
       ORT_WAVES5=0 / 1 forces. */
    const bool can_five = !wavefront && !counters && tabs && rv.mode != JOBS_EXPLICIT && kn.wide <= 0 && kn.exchange <= 0 &&
                          ort_w5_sizeof_scene_view() == sizeof(SceneView) && ort_w5_sizeof_render_hot() == sizeof(RenderHot);
    /* persistent grid: 4 blocks of 256 lanes per CU (5 for the five-waves kernels, decided below), never more lanes than jobs */
    unsigned long long lanes_wanted = rv.job_count;
    unsigned int max_blocks = d->max_blocks;
    unsigned int grid = (unsigned int)((lanes_wanted + kBlock - 1) / kBlock);
    if (grid > max_blocks) grid = max_blocks;
    if (grid == 0) grid = 1;
    bool exch = false;
    if (!wavefront) {
        /* ray exchange (pt_lane_x): bit-identical; 60 of 64 lanes in the shading pass instead of 53 and leaf
           visits four times better filled, against the parking traffic.  On by itself where it is a gain
           (profiles/r02_tuning.md): the diffuse flavour (the all-lobes one spills too much around the exchange) on
           launches of at least 24 jobs per lane -- every parked path is a job in progress, so a wave's tail grows with
           what it has parked, which short launches cannot amortise (round 3, stashes drained over the last four jobs per
           lane: 2- / 4- / 8-way shard of the headline frame, 63 / 32 / 16 jobs per lane: 216.2 / 115.5 / 65.2 ms with the exchange,
           228.6 / 117.0 / 63.5 plain).  ORT_EXCHANGE=0 / 1 forces it. */
        /* ... and not for trees that leave the L2: the 1M-triangle scene runs 1 392 Mpaths/s with it and 1 393 without (round 2:
           1 268 / 1 272), and its stashes would move 3 TB/s through the fabric for that */
        /* ... and only where rays spend their time in the tree: since a wave draws its jobs in batches of like jobs (draw_job) the
           plain loop keeps its lanes together by itself, and the exchange pays from an SAH cost of the tree (expected node visits of a
           random ray through the scene box, ort_tree.cpp) of about 0.09 -- 1080p / 512 spp, exchange / plain loop, Mpaths/s: bunny at
           scale 3 / 5 / 8 / 12 (SAH cost 0.027 / 0.076 / 0.19 / 0.44) 5 428 / 5 844, 5 143 / 5 151, 3 904 / 3 728, 3 173 / 2 946; dwarf at
           scale 0.008 / 0.012 / 0.02 / 0.03 (0.018 / 0.040 / 0.11 / 0.25) 5 335 / 5 666, 4 921 / 5 143, 4 454 / 4 276, 3 626 / 3 483 */
        const bool worth_it = diffuse && cache_resident_tree && scene->tree.sah_cost >= 0.09f && rv.job_count >= 24ull * (unsigned long long)grid * kBlock;
        exch = tabs && rv.mode != JOBS_EXPLICIT && (kn.exchange >= 0 ? kn.exchange != 0 : worth_it) && (!counters || (want_util && diffuse));
        if (exch && kn.refill_below < 0) rv.refill_below = 24; /* stragglers park instead of idling: leave the loop a little earlier (dwarf room 4K, 16 / 24 / 48: 4 466 / 4 536 / 4 536 Mpaths/s) */
        if (exch) {
            rv.capL = kCapL; rv.capR = kCapR;
            rv.long_min = kn.long_min >= 0 ? (uint32_t)kn.long_min : 64u;
            rv.long_refill = kn.long_refill >= 0 ? (uint32_t)kn.long_refill : 32u;
            rv.inflight_cap = kn.inflight_cap >= 0 ? (uint32_t)kn.inflight_cap : 64u;
            rv.park_min = kn.park_min >= 0 ? (uint32_t)kn.park_min : 1u;
            if (rv.long_min < 1u) rv.long_min = 1u;
            if (rv.long_min > rv.capL) rv.long_min = rv.capL;
            if (rv.long_refill > 64u) rv.long_refill = 64u;
            /* a wave whose lanes all hold off new jobs (parked paths >= inflight_cap) must be able to start a traversal phase
               on what it has parked (parked + tracing >= long_min), or nothing in it could ever move again */
            if (rv.inflight_cap < rv.long_min) rv.inflight_cap = rv.long_min;
            if (rv.inflight_cap < 1u) rv.inflight_cap = 1u;
            {
                /* the last FOUR jobs per lane (round 3, whole frame / 2- / 4- / 8-way shard: 0 jobs 431.5 / 224.4 / 122.3 / 71.6 ms, two
                   426.3 / 219.8 / 117.5 / 66.1, four 426.4 / 218.8 / 116.6 / 64.6) */
                const unsigned long long quarter_jobs = kn.endgame_jobs >= 0 ? (unsigned long long)kn.endgame_jobs : 16ull;
                const unsigned long long tail_jobs = quarter_jobs * (unsigned long long)grid * kBlock / 4ull;
                rv.endgame_from = rv.job_count > tail_jobs ? rv.job_count - tail_jobs : 0ull;
            }
            rv.stash_wave_f4 = (kStashVecs + (uint32_t)kLdsStack / 4u) * rv.capL + kStashVecs * rv.capR;
            const size_t need = (size_t)d->max_blocks * (kBlock / 64) * rv.stash_wave_f4 * sizeof(float4);
            if ((rc = ensure(&d->stash, &d->stash_bytes, need, err))) return rc;
            rv.stash = (float4 *)d->stash;
        }
    }
    const bool five = can_five && !exch && (kn.waves5 >= 0 ? kn.waves5 != 0 : (!diffuse || !cache_resident_tree));
    if (five && kn.blocks_per_cu <= 0) {
        max_blocks = (unsigned int)(d->cu_count > 0 ? d->cu_count : 256) * 5u;
        grid = (unsigned int)((lanes_wanted + kBlock - 1) / kBlock);
        if (grid > max_blocks) grid = max_blocks;
        if (grid == 0) grid = 1;
    }
    /* CHUNK renders issue their jobs block-major (see "the order in which a CHUNK render issues its jobs"); ORT_LPT=0:
       chunk-major as in rounds 1-2 (A/B runs; same image either way) */
    if (rv.mode == JOBS_CHUNK && rv.nchunks >= 2u && kn.lpt != 0) rv.block_major = 1u;
    /* a wave draws its job indices in batches (ort_lane.h: draw_job) until ORT_BATCH_TAIL jobs per lane are left in the job
       space, then one by one: the end of a launch is dealt as finely as before */
    if (!wavefront) {
        /* 64 indices at a time, 128 on launches of 96 jobs per lane and more; what a wave holds back is at most two jobs per lane of
           its own, of up to eight average job lengths each in the expensive blocks: batches stop 8 (16) jobs per lane before the
           end.  Whole headline frame / its 8-way shard, ms: no batches 422.5 / 63.7, 32: 418.9 / 63.1, 64: 414.3 / 62.0, 128: 411.3 /
           74.4 (with the tail of 64), 256: 419.9 / 113 (profiles/r03_tuning.md) */
        const unsigned long long lanes = (unsigned long long)grid * kBlock;
        rv.job_batch = kn.job_batch >= 0 ? (uint32_t)kn.job_batch : (rv.job_count >= 96ull * lanes ? 128u : 64u);
        const unsigned long long per_lane = kn.batch_tail >= 0 ? (unsigned long long)kn.batch_tail : 8ull * ((rv.job_batch + 63u) / 64u);
        const unsigned long long tail = per_lane * lanes;
        rv.batch_until = rv.job_count > tail ? rv.job_count - tail : 0ull;
    }
    if (kn.debug_drain && stats && !wavefront) {
        const size_t bytes = (size_t)max_blocks * (kBlock / 64) * sizeof(unsigned long long);
        if ((rc = ensure(&d->drain, &d->drain_bytes, bytes, err))) return rc;
        ORT_HIP(hipMemsetAsync(d->drain, 0, bytes, stream));
        rv.drain = (unsigned long long *)d->drain;
    }
    /* the RenderView goes to HBM (pageable source: the copy is staged before the call returns); the kernels get the few
       fields every ray reads by value and a pointer to the rest */
    ORT_HIP(hipMemcpyAsync(d->rv_dev, &rv, sizeof(RenderView), hipMemcpyHostToDevice, stream));
    RenderHot hot{};
    hot.mode = rv.mode; hot.W = rv.W; hot.H = rv.H; hot.rr = rv.rr;
    hot.refill_below = rv.refill_below; hot.descend_below = rv.descend_below;
    hot.c = (const ORT_CONSTANT_AS RenderView *)d->rv_dev;
    if (stats) ORT_HIP(hipEventRecord(d->ev0, stream));
    if (wavefront) {
        rc = counters ? launch_wavefront<true>(d, sv, rv, hot, stream, err) : launch_wavefront<false>(d, sv, rv, hot, stream, err);
        if (rc) return rc;
    } else {
#define ORT_LAUNCH(C, D, T) hipLaunchKernelGGL((pt_persistent<C, D, T>), dim3(grid), dim3(kBlock), 0, stream, sv, hot)
        /* 4-wide tree (DevNode4): half the dependent node fetches per ray -- and twice the vector instructions per visit, in a
           kernel that is issue-bound at a third of its lanes on the trees it was meant for: 1 218 against 1 368 Mpaths/s on the
           1M-triangle scene (profiles/r03_tuning.md).  Off unless ORT_WIDE=1 asks for it (same image either way). */
        const bool wide = d->nodes4 && !exch && tabs && (counters || rv.mode != JOBS_EXPLICIT) && !(counters && diffuse && want_util) && kn.wide > 0;
        if (five && !exch) {
            ort_launch_w5(diffuse ? 1 : 0, grid, (void *)stream, &sv, &hot);
        } else
        if (wide) {
            sv.nodes = (const float4 *)d->nodes4;
            if (counters) hipLaunchKernelGGL((pt_persistent<true, false, true, false, true>), dim3(grid), dim3(kBlock), 0, stream, sv, hot);
            else if (diffuse) hipLaunchKernelGGL((pt_persistent<false, true, true, true, true>), dim3(grid), dim3(kBlock), 0, stream, sv, hot);
            else hipLaunchKernelGGL((pt_persistent<false, false, true, true, true>), dim3(grid), dim3(kBlock), 0, stream, sv, hot);
        } else
        if (exch) {
            if (counters) hipLaunchKernelGGL((pt_persistent_x<true, true>), dim3(grid), dim3(kBlock), 0, stream, sv, hot); /* diagnostics: probes of the diffuse flavour */
            else if (diffuse) hipLaunchKernelGGL((pt_persistent_x<false, true>), dim3(grid), dim3(kBlock), 0, stream, sv, hot);
            else hipLaunchKernelGGL((pt_persistent_x<false, false>), dim3(grid), dim3(kBlock), 0, stream, sv, hot);
        } else
        if (counters && diffuse && want_util) { if (tabs) ORT_LAUNCH(true, true, true); else ORT_LAUNCH(true, true, false); }
        else if (counters) { if (tabs) ORT_LAUNCH(true, false, true); else ORT_LAUNCH(true, false, false); }
        /* IMPLICIT job spaces (PIXEL / CHUNK policies): the variant whose lanes carry no job rect / count / index */
        else if (diffuse) { if (tabs && rv.mode != JOBS_EXPLICIT) hipLaunchKernelGGL((pt_persistent<false, true, true, true>), dim3(grid), dim3(kBlock), 0, stream, sv, hot);
                            else if (tabs) ORT_LAUNCH(false, true, true); else ORT_LAUNCH(false, true, false); }
        else { if (tabs && rv.mode != JOBS_EXPLICIT) hipLaunchKernelGGL((pt_persistent<false, false, true, true>), dim3(grid), dim3(kBlock), 0, stream, sv, hot);
               else if (tabs) ORT_LAUNCH(false, false, true); else ORT_LAUNCH(false, false, false); }
#undef ORT_LAUNCH
        ORT_HIP(hipGetLastError());
    }
    if (rv.mode == JOBS_CHUNK) {
        unsigned long long total = (unsigned long long)rv.my_blocks * 64ull;
        unsigned int cgrid = (unsigned int)((total + 255) / 256);
        if (cgrid) hipLaunchKernelGGL(combine_chunks, dim3(cgrid), dim3(256), 0, stream, hot);
        ORT_HIP(hipGetLastError());
    }
    if (stats) ORT_HIP(hipEventRecord(d->ev1, stream));

    if (!d_out && h_out) ORT_HIP(hipMemcpyAsync(h_out, out, image_bytes, hipMemcpyDeviceToHost, stream));
    if (final_states) ORT_HIP(hipMemcpyAsync(final_states, d->states, (size_t)job_count * 4u, hipMemcpyDeviceToHost, stream));
    /* a fallback queue overflow (cannot happen by construction; tripwire) must not go unnoticed: every synchronous form of the call checks it
       (the fire-and-forget device form, stats == NULL, cannot without a sync; bench.py asks for stats) */
    ORT_HIP(hipEventRecord(d->ev_done, stream));
    d->inflight = true;
    if (stats || !d_out || final_states) {
        ORT_HIP(hipStreamSynchronize(stream));
        d->inflight = false;
        unsigned long long ovf = 0;
        ORT_HIP(hipMemcpy(&ovf, d->ctrl + 7, sizeof(ovf), hipMemcpyDeviceToHost));
        if (ovf) { *err = "reference-order fallback queue overflowed"; return ORT_ERR_UNSUPPORTED; }
    }
    if (stats && kn.debug_fallback) { /* developer diagnostics */
        unsigned long long fb[2], dg[3];
        ORT_HIP(hipMemcpy(fb, d->ctrl + 6, sizeof(fb), hipMemcpyDeviceToHost));
        ORT_HIP(hipMemcpy(dg, d->ctrl + 6 + kDiagFallback, sizeof(dg), hipMemcpyDeviceToHost));
        fprintf(stderr, "fallback: %llu rays traversed again without their first winner, %llu re-cast exactly (%llu octree nodes enqueued, %llu busy queues met)\n",
                dg[1], fb[0], dg[0], dg[2]);
        fprintf(stderr, "issue order: %s\n", rv.block_major ? "block-major" : "chunk-major");
    }
    if (stats && rv.drain) { /* developer diagnostics: how the launch drains */
        std::vector<unsigned long long> t((size_t)grid * (kBlock / 64));
        ORT_HIP(hipMemcpy(t.data(), rv.drain, t.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        unsigned long long last = 0;
        for (unsigned long long v : t) last = v > last ? v : last;
        const double tick_ms = 1e-5; /* s_memrealtime: 100 MHz */
        const double marks[] = {0.25, 0.5, 1, 2, 3, 5, 8, 12, 20};
        fprintf(stderr, "drain: of %zu waves, still running before the end of the launch:", t.size());
        for (double m : marks) {
            size_t n = 0;
            for (unsigned long long v : t) n += (double)(last - v) * tick_ms < m ? 1 : 0;
            fprintf(stderr, "  %g ms: %zu", m, n);
        }
        fprintf(stderr, "\n");
    }
    if (stats) {
        memset(stats, 0, sizeof(*stats));
        float ms = 0;
        ORT_HIP(hipEventElapsedTime(&ms, d->ev0, d->ev1));
        stats->kernel_ms = ms;
        {
            unsigned long long c[6];
            ORT_HIP(hipMemcpy(c, d->ctrl + 1, sizeof(c), hipMemcpyDeviceToHost));
            stats->fallback_rays = c[5]; /* counted by every kernel flavour (straight to memory, rare) */
            if (counters) { stats->paths = c[0]; stats->rays = c[1]; stats->node_tests = c[2]; stats->tri_tests = c[3]; stats->analytic_tests = c[4]; }
        }
        if (counters) {
            if (want_util) {
                static const char *names[8] = {"node visit", "leaf visit", "traverse outer iteration", "shade call", "  of which lanes with a finished ray",
                                               "produce_ray pass", "  bounce draw", "  sin/cos + ray setup"};
                unsigned long long u[16];
                ORT_HIP(hipMemcpy(u, d->ctrl + 8, sizeof(u), hipMemcpyDeviceToHost));
                for (int k = 0; k < 8; ++k)
                    fprintf(stderr, "util %-40s wave-events %14llu  mean active lanes %6.2f\n", names[k], u[2 * k],
                            u[2 * k] ? (double)u[2 * k + 1] / (double)u[2 * k] : 0.0);
                static const char *pnames[10] = {"resolve_hit (chain check)", "hit processing + bounce draw", "pixel / job / new sample", "sin/cos + normalise + ray setup",
                                                "1/d + analytic prologue", "descend loop", "leaf", "pt_lane loop top (after traverse)",
                                                "produce_ray entry (after resolve)", "traverse loop top"};
                unsigned long long ph[30];
                ORT_HIP(hipMemcpy(ph, d->ctrl + 8 + 32, sizeof(ph), hipMemcpyDeviceToHost));
                double total = 0;
                for (int k = 0; k < 10; ++k) total += (double)ph[3 * k];
                for (int k = 0; k < 10; ++k)
                    fprintf(stderr, "phase %-36s share %5.1f %%  cycles/mark %8.1f  marks %12llu  lanes at mark %5.1f\n", pnames[k],
                            total > 0 ? 100.0 * (double)ph[3 * k] / total : 0.0, ph[3 * k + 1] ? (double)ph[3 * k] / (double)ph[3 * k + 1] : 0.0,
                            ph[3 * k + 1], ph[3 * k + 1] ? (double)ph[3 * k + 2] / (double)ph[3 * k + 1] : 0.0);
            }
        }
    }
    return ORT_OK;
}

} // namespace ort

#endif /* !ORT_HOST_SIM */
