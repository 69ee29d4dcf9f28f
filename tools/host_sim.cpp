/*
 * tools/host_sim.cpp -- DEVELOPER HARNESS, not part of the product: the kernel's lane code on host threads.
 *
 * The dev container has no GPU.  This tool compiles the kernel's lane function
 * (ort_lane.h: pt_lane) as ordinary host C++ (-DORT_HOST_SIM) and runs one simulated
 * lane per host thread (SIM_THREADS, default 1) over the job space, so kernel logic can be debugged
 * against the oracle before spending GPU-box time.  With policy tile32 and SIM_THREADS = cores it is
 * the Linux counterpart of the reference's own driver: main()'s 1 024 tiles
 * (macos_main.mm:602-662) handed out by one job counter to a pool of worker threads
 * (macos_main.mm:565-598 starts eight pthreads on a work queue), every worker a lane of the
 * kernel.  The library never loads, links or runs it: the render call has no CPU fallback
 * (tests/test_host.py holds its image against the oracle; nothing else uses it).
 *
 * build: see tools/Makefile     run: [SIM_THREADS=n] host_sim <scn> <base> W H spp seed policy chunk out.f32
 */
#define ORT_HOST_SIM 1
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
static uint32_t *g_pixel_rng; static int g_W;
static int g_dbg_x = -1, g_dbg_y = -1; static FILE *g_ray_log;
#define ORT_SIM_RAY_HOOK(PX_, PY_, O_, D_, T_, N_, M_) do { if ((PX_) == g_dbg_x && (PY_) == g_dbg_y && g_ray_log) { float r_[11] = {O_.x, O_.y, O_.z, D_.x, D_.y, D_.z, T_, N_.x, N_.y, N_.z, 0}; unsigned m_ = (M_); fwrite(r_, 4, 10, g_ray_log); fwrite(&m_, 4, 1, g_ray_log); } } while (0)
#define ORT_SIM_PIXEL_HOOK(x, y, rng) do { if (g_pixel_rng) g_pixel_rng[(y) * g_W + (x)] = (rng); } while (0)
#include "../offline_raytracer_amd/csrc/ort_lane.h"

#include <algorithm>
#include <chrono>
#include <thread>

using namespace ort;
#ifdef ORT_CHAIN_STATS
namespace ort { unsigned long long g_cs[4][16]; }
static void cs_dump() { static const char *nm[4] = {"chain len:", "first-outside depth from top (15 = none):", "re-cast reasons [1 phantom first, 2 unknown + hit in gap, 6 unknown + phantom in gap, 3 reject + phantom, 4 reject + second verdict]:", "first verdict [0 reject, 1 unknown; 4 + kind of W]:"};
    for (int k = 0; k < 4; ++k) { fprintf(stderr, "%s", nm[k]); for (int i = 0; i < 16; ++i) fprintf(stderr, " %llu", g_cs[k][i]); fprintf(stderr, "\n"); } }
#endif

int main(int argc, char **argv) {
    if (argc < 10) { fprintf(stderr, "usage: host_sim scn base W H spp seed policy chunk out.f32 [shard_index shard_count]\n"); return 2; }
    int W = atoi(argv[3]), H = atoi(argv[4]);
    uint32_t spp = (uint32_t)strtoul(argv[5], 0, 10), seed = (uint32_t)strtoul(argv[6], 0, 10);
    std::string policy = argv[7];
    uint32_t chunk = (uint32_t)strtoul(argv[8], 0, 10);
    ort_scene *scene = nullptr;
    if (ort_scene_load_scn(argv[1], argv[2], &scene) != ORT_OK || ort_scene_commit(scene) != ORT_OK) {
        fprintf(stderr, "scene: %s\n", ort_last_error());
        return 1;
    }
    const Tree &t = scene->tree;
    ort_tree_info ti;
    ort_scene_get_tree_info(scene, &ti);
    fprintf(stderr, "tree: %u nodes, %u leaves, max leaf %u, depth %u, sah %.2f\n", ti.node_count, ti.leaf_count, ti.max_leaf_prims, ti.max_depth, ti.sah_cost);

    std::vector<DevMaterial> mats(scene->materials.size());
    for (size_t i = 0; i < mats.size(); ++i) mats[i] = make_dev_material(scene->materials[i]);
    std::vector<uint32_t> lis(scene->lights.size());
    for (size_t i = 0; i < lis.size(); ++i) lis[i] = scene->lights[i].type == 1u;

    SceneView sv{};
    sv.nodes = (const float4 *)t.nodes.data(); sv.tris = (const float4 *)t.tris.data();
    sv.spheres = (const float4 *)t.spheres.data(); sv.boxes = (const float4 *)t.boxes.data(); sv.cyls = (const float4 *)t.cyls.data();
    static std::vector<PrimInfo> prim_info;
    build_prim_info(t, scene->ref, prim_info, sv.info_box, sv.info_cyl, sv.info_sphere);
    sv.prim_info = prim_info.data();
    sv.materials = (const float4 *)mats.data();
    sv.light_is_sphere = lis.data(); sv.light_count = (uint32_t)lis.size();
    sv.pro_boxes = t.pro_boxes; sv.pro_spheres = t.pro_spheres; sv.pro_cyls = t.pro_cyls;
    const RefTree &rt = scene->ref;
    static SceneCold cold; sv.cold = &cold;
    cold.ref_nodes = (const float4 *)rt.nodes.data(); cold.ref_recs = rt.recs.data(); sv.chain_boxes = (const float4 *)rt.chain_boxes.data();
    cold.tri_order = rt.tri_order.data(); cold.sphere_order = rt.sphere_order.data(); cold.box_order = rt.box_order.data(); cold.cyl_order = rt.cyl_order.data();
    fprintf(stderr, "ref octree: %zu nodes, %u leaves, max leaf %u, chain boxes %zu\n", rt.nodes.size(), rt.nonempty_leaves, rt.max_leaf_records, rt.chain_boxes.size() / 2);
    ort_camera cam;
    camera_basis(*scene, W, H, &cam);
    memcpy(sv.cam, &cam, sizeof(cam));

    std::vector<float> out((size_t)W * H * 3, 0.0f), partial;
    unsigned long long ctrl[8] = {0};
    RenderView rv{};
    rv.W = W; rv.H = H; rv.x0 = 0; rv.y0 = 0; rv.x1 = W; rv.y1 = H;
    rv.seed = seed; rv.spp = spp; rv.chunk = chunk; rv.rr = getenv("SIM_RR") ? (float)atof(getenv("SIM_RR")) : 0.8f;
    rv.refill_below = 12;
    rv.descend_below = getenv("SIM_DESCEND_BELOW") ? atoi(getenv("SIM_DESCEND_BELOW")) : 8;
    rv.out = out.data(); rv.next_job = ctrl; rv.counters = ctrl + 1;
    cold.fallback_counters = ctrl + 6;
    rv.shard_count = argc > 11 ? (uint32_t)atoi(argv[11]) : 1; rv.shard_index = argc > 11 ? (uint32_t)atoi(argv[10]) : 0;
    rv.block_x0 = rv.block_y0 = 0;
    rv.blocks_w = (uint32_t)((W + 7) / 8);
    uint32_t blocks_total = rv.blocks_w * (uint32_t)((H + 7) / 8);
    rv.my_blocks = (blocks_total - rv.shard_index + rv.shard_count - 1) / rv.shard_count;
    std::vector<ort_tile_job> jobs;
    std::vector<uint32_t> finals;
    if (policy == "pixel") { rv.mode = JOBS_PIXEL; rv.nchunks = 1; rv.job_count = (unsigned long long)rv.my_blocks * 64; }
    else if (policy == "chunk") {
        rv.mode = JOBS_CHUNK; rv.nchunks = spp / chunk; rv.job_count = (unsigned long long)rv.my_blocks * 64 * rv.nchunks;
        partial.assign((size_t)rv.nchunks * rv.my_blocks * 64 * 3, 0.0f); rv.partial = partial.data(); /* packed block layout: edge blocks are whole */
    } else {
        uint32_t master = seed;
        auto xs = [&]() { master ^= master << 13; master ^= master >> 17; master ^= master >> 5; return master; };
        if (policy == "whole") jobs.push_back(ort_tile_job{0, 0, W, H, xs(), spp});
        else {
            int tw = (int)ceilf(W / 32.0f), th = (int)ceilf(H / 32.0f);
            for (int ty = 0; ty < 32; ++ty) for (int tx = 0; tx < 32; ++tx) {
                ort_tile_job j{tx * tw, ty * th, std::min(W, tx * tw + tw), std::min(H, ty * th + th), xs(), spp};
                if (j.x0 < j.x1 && j.y0 < j.y1) jobs.push_back(j);
            }
        }
        finals.resize(jobs.size());
        rv.mode = JOBS_EXPLICIT; rv.jobs = jobs.data(); rv.job_count = jobs.size(); rv.final_states = finals.data();
    }
    std::vector<uint32_t> pix_rng((size_t)W * H, 0);
    RenderHot hot{};
    hot.mode = rv.mode; hot.W = rv.W; hot.H = rv.H; hot.rr = rv.rr; hot.refill_below = rv.refill_below; hot.descend_below = rv.descend_below;
    hot.c = &rv;
    if (getenv("SIM_RAY_LOG")) { g_ray_log = fopen(getenv("SIM_RAY_LOG"), "wb"); g_dbg_x = atoi(getenv("SIM_X")); g_dbg_y = atoi(getenv("SIM_Y")); }
    if (getenv("SIM_DUMP_RNG")) { g_pixel_rng = pix_rng.data(); g_W = W; }
    std::vector<uint32_t> lds(kLdsStack * kBlock);
    auto t0 = std::chrono::steady_clock::now();
    std::vector<float> lds_focal(3 * kBlock);
    std::vector<uint32_t> bfsq(scene->ref.nodes.size() + 8), bfs_lock(1, 0u);
    cold.bfs_pool = bfsq.data();
    cold.bfs_locks = bfs_lock.data();
    cold.bfs_queue_cap = (uint32_t)bfsq.size();
    cold.bfs_queue_count = 1;
    sv.force_fallback_mask = getenv("SIM_FORCE_FALLBACK") ? (uint32_t)strtoul(getenv("SIM_FORCE_FALLBACK"), 0, 0) : 0xffffffffu;
    const int n_threads = getenv("SIM_THREADS") ? std::max(1, atoi(getenv("SIM_THREADS"))) : 1;
    if (getenv("SIM_WAVEFRONT")) {
        /* the wavefront schedule with a small slot pool: shade all slots, trace all slots, repeat */
        uint32_t S = (uint32_t)atoi(getenv("SIM_WAVEFRONT"));
        if (S > rv.job_count) S = (uint32_t)rv.job_count;
        std::vector<float4> od0(S), hit0(S), p0(S), p1(S), p2(S);
        std::vector<float2> od1(S);
        std::vector<uint4> p3(S);
        std::vector<uint32_t> hitp(S), flags(S, (uint32_t)PS_NEED_JOB);
        unsigned long long active = 0;
        WfView wf{S, od0.data(), od1.data(), hit0.data(), hitp.data(), p0.data(), p1.data(), p2.data(), p3.data(), flags.data(), &active};
        std::vector<uint32_t> wlds(kWfLdsStack * kBlock), wspill(kWfSpill);
        Counters c;
        for (;;) {
            unsigned long long produced = 0;
            for (uint32_t i = 0; i < S; ++i) produced += wf_shade_slot<true>(sv, hot, nullptr, wf, i, c) ? 1 : 0;
            if (!produced) break;
            for (uint32_t i = 0; i < S; ++i) wf_trace_slot<true>(sv, nullptr, wf, i, 0, wlds.data(), wspill.data(), 0, c);
        }
        flush_counters(hot, c, true);
    } else
    if (n_threads > 1) {
        /* a pool of workers on one job counter: every worker owns what a GPU lane owns (traversal stack, focal-point cache, its
           queue of the exact fallback), shares what the lanes share (scene, job counter, work counters, framebuffer) */
        const bool wide = getenv("SIM_WIDE") != nullptr, diffuse_only = getenv("SIM_DIFFUSE") != nullptr;
        if (wide && t.nodes4.empty()) { fprintf(stderr, "no wide tree\n"); return 1; }
        std::vector<std::thread> pool;
        for (int w = 0; w < n_threads; ++w)
            pool.emplace_back([&, w]() {
                SceneView svw = sv;
                SceneCold coldw = cold;
                std::vector<uint32_t> q(scene->ref.nodes.size() + 8), lock(1, 0u), stack(kLdsStack * kBlock);
                std::vector<float> focal(3 * kBlock);
                coldw.bfs_pool = q.data(); coldw.bfs_locks = lock.data(); coldw.bfs_queue_cap = (uint32_t)q.size(); coldw.bfs_queue_count = 1;
                svw.cold = &coldw;
                if (wide) { svw.nodes = (const float4 *)t.nodes4.data(); pt_lane<true, false, false, false, true>(svw, hot, nullptr, stack.data(), focal.data(), 0, (uint32_t)w); }
                else if (diffuse_only) pt_lane<true, true>(svw, hot, nullptr, stack.data(), focal.data(), 0, (uint32_t)w);
                else pt_lane<true>(svw, hot, nullptr, stack.data(), focal.data(), 0, (uint32_t)w);
            });
        for (auto &th : pool) th.join();
    } else
    if (getenv("SIM_WIDE")) { /* the 4-wide form of the tree (DevNode4, visit_node4) */
        if (t.nodes4.empty()) { fprintf(stderr, "no wide tree\n"); return 1; }
        fprintf(stderr, "wide tree: %zu nodes, depth %u\n", t.nodes4.size(), t.max_depth4);
        sv.nodes = (const float4 *)t.nodes4.data();
        pt_lane<true, false, false, false, true>(sv, hot, nullptr, lds.data(), lds_focal.data(), 0, 0);
    } else
    if (getenv("SIM_DIFFUSE")) pt_lane<true, true>(sv, hot, nullptr, lds.data(), lds_focal.data(), 0, 0); /* caller vouches for Ks = Kt = 0 */
    else pt_lane<true>(sv, hot, nullptr, lds.data(), lds_focal.data(), 0, 0); /* TABS = false: the small tables are read from their arrays */
    if (rv.mode == JOBS_CHUNK)
        for (unsigned long long i = 0; i < (unsigned long long)rv.my_blocks * 64; ++i) combine_pixel(hot, i);
    double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    
#ifdef ORT_CHAIN_STATS
    cs_dump();
#endif
    fprintf(stderr, "sim: %.2fs paths %llu rays %llu node_tests %llu tri_tests %llu analytic %llu fallback %llu overflow %llu%s\n", sec, ctrl[1], ctrl[2], ctrl[3], ctrl[4], ctrl[5], ctrl[6], ctrl[7],
            finals.empty() ? "" : (" final_rng " + std::to_string(finals.back())).c_str());
#ifdef ORT_CHAIN_CROSSCHECK
    fprintf(stderr, "chain shortcut == full walk on %llu rays; unnested chains %u\n", g_chain_crosschecks, scene->ref.unnested_chains);
#endif
    if (g_ray_log) fclose(g_ray_log);
    if (g_pixel_rng) { FILE *g = fopen(getenv("SIM_DUMP_RNG"), "wb"); fwrite(pix_rng.data(), 4, pix_rng.size(), g); fclose(g); }
    FILE *f = fopen(argv[9], "wb");
    fwrite(out.data(), 4, out.size(), f);
    fclose(f);
    return 0;
}
