/*
 * ort_render -- command-line driver: the Linux/MI355X replacement for the reference's
 * main() (code/macos_main.mm:289-710), with its literals turned into flags: scene path
 * (:317), resolution (:319-320), samples per pixel (:612), seed (:297-298), roulette (:656).
 * Loads a .scn, builds the tree, uploads to a GPU, renders through the C ABI, writes .hdr.
 * --gpus N: the reference's eight-thread tile pool (macos_main.mm:565-671) across N GPUs of the node, from this one
 * process: the scene is loaded once per device, every device renders its 8x8 blocks (block id % N) into a packed
 * buffer on its own host thread, and one RCCL gather (ort_gather_framebuffer_local) assembles the frame on device 0.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <chrono>
#include <string>
#include <thread>
#include <vector>

#include <hip/hip_runtime_api.h> /* device buffers of the --gpus N path (host API only) */

#include "../../include/ort.h"

static void usage(const char *argv0) {
    fprintf(stderr,
            "usage: %s --scene file.scn [--base dir/] [--width W --height H] [--spp N] [--seed S]\n"
            "          [--policy tile32|whole|pixel|chunk] [--chunk C] [--rr 0.8] [--device D | --gpus N]\n"
            "          [--out image.hdr] [--raw image.f32]\n",
            argv0);
}

int main(int argc, char **argv) {
    std::string scene_path, base, out_path = "output.hdr", raw_path, policy = "chunk";
    int width = 0, height = 0, device = 0, gpus = 1;
    unsigned spp = 64, seed = 12345, chunk = 0;
    float rr = 0.8f;
    for (int i = 1; i < argc; ++i) {
        std::string a = argv[i];
        auto next = [&](const char *what) -> const char * {
            if (i + 1 >= argc) { fprintf(stderr, "missing value for %s\n", what); exit(2); }
            return argv[++i];
        };
        if (a == "--scene") scene_path = next("--scene");
        else if (a == "--base") base = next("--base");
        else if (a == "--width") width = atoi(next("--width"));
        else if (a == "--height") height = atoi(next("--height"));
        else if (a == "--spp") spp = (unsigned)strtoul(next("--spp"), 0, 10);
        else if (a == "--seed") seed = (unsigned)strtoul(next("--seed"), 0, 10);
        else if (a == "--chunk") chunk = (unsigned)strtoul(next("--chunk"), 0, 10);
        else if (a == "--policy") policy = next("--policy");
        else if (a == "--rr") rr = (float)atof(next("--rr"));
        else if (a == "--device") device = atoi(next("--device"));
        else if (a == "--gpus") gpus = atoi(next("--gpus"));
        else if (a == "--out") out_path = next("--out");
        else if (a == "--raw") raw_path = next("--raw");
        else { usage(argv[0]); return 2; }
    }
    if (scene_path.empty()) { usage(argv[0]); return 2; }
    if (base.empty()) {
        size_t slash = scene_path.find_last_of('/');
        base = (slash == std::string::npos) ? std::string("") : scene_path.substr(0, slash + 1);
    }
    ort_scene *scene = nullptr;
    if (ort_scene_load_scn(scene_path.c_str(), base.c_str(), &scene) != ORT_OK) {
        fprintf(stderr, "load failed: %s\n", ort_last_error());
        return 1;
    }
    ort_scene_info info;
    ort_scene_get_info(scene, &info);
    if (width <= 0) width = info.screen_width > 0 ? info.screen_width : 1280;   /* main() forces 1280x720 */
    if (height <= 0) height = info.screen_height > 0 ? info.screen_height : 720;
    if (gpus > 1 && (policy == "tile32" || policy == "whole")) {
        fprintf(stderr, "--gpus N needs a per-pixel seeding policy (pixel or chunk): main()'s serial tile streams do not shard by blocks\n");
        return 2;
    }
    if (ort_scene_commit(scene) != ORT_OK || (gpus <= 1 && ort_scene_upload(scene, device) != ORT_OK)) {
        fprintf(stderr, "scene setup failed: %s\n", ort_last_error());
        return 1;
    }
    ort_render_params p{};
    p.width = width; p.height = height; p.x0 = 0; p.y0 = 0; p.x1 = width; p.y1 = height;
    p.seed = seed; p.spp = spp; p.rr = rr;
    if (policy == "tile32") p.policy = ORT_POLICY_TILE32;
    else if (policy == "whole") p.policy = ORT_POLICY_WHOLE;
    else if (policy == "pixel") p.policy = ORT_POLICY_PIXEL;
    else p.policy = ORT_POLICY_CHUNK;
    if (p.policy == ORT_POLICY_CHUNK) {
        if (chunk == 0) { chunk = spp; while (chunk > 64 && chunk % 2 == 0) chunk /= 2; }
        p.chunk = chunk;
    }
    std::vector<float> image((size_t)width * height * 3, 0.0f);
    ort_stats st{};
    auto t0 = std::chrono::steady_clock::now();
    if (gpus > 1) {
        int have = 0;
        /* ORT_CLI_SHARE_DEVICE=1: all shards on device 0 (a rehearsal of the sharded path on a one-GPU box) */
        const bool share = getenv("ORT_CLI_SHARE_DEVICE") != nullptr;
        if (ort_device_count(&have) != ORT_OK || (have < gpus && !share)) { fprintf(stderr, "--gpus %d: only %d HIP device(s) visible\n", gpus, have); return 1; }
        std::vector<ort_scene *> scenes((size_t)gpus, nullptr);
        std::vector<void *> packed((size_t)gpus, nullptr);
        std::vector<int> devices((size_t)gpus), rcs((size_t)gpus, ORT_OK);
        std::vector<std::string> errs((size_t)gpus);
        std::vector<ort_stats> stats((size_t)gpus);
        std::vector<std::thread> workers;
        for (int r = 0; r < gpus; ++r) {
            devices[(size_t)r] = share ? 0 : r;
            workers.emplace_back([&, r]() {
                const int dev = devices[(size_t)r];
                /* one host thread per device: its own copy of the scene (loading is cheap next to rendering) and its blocks */
                ort_scene *s = r == 0 ? scene : nullptr;
                int rc = ORT_OK;
                if (r != 0) rc = ort_scene_load_scn(scene_path.c_str(), base.c_str(), &s);
                if (rc == ORT_OK && r != 0) rc = ort_scene_commit(s);
                if (rc == ORT_OK) rc = ort_scene_upload(s, dev);
                scenes[(size_t)r] = s;
                ort_render_params q = p;
                q.flags |= ORT_RENDER_PACKED;
                q.shard_index = (uint32_t)r; q.shard_count = (uint32_t)gpus;
                uint64_t nb = 0;
                if (rc == ORT_OK) rc = ort_shard_block_count(width, height, (uint32_t)r, (uint32_t)gpus, &nb);
                if (rc == ORT_OK && (hipSetDevice(dev) != hipSuccess || hipMalloc(&packed[(size_t)r], (size_t)(nb ? nb : 1) * 768u) != hipSuccess)) {
                    errs[(size_t)r] = "hipMalloc of the packed framebuffer failed"; rcs[(size_t)r] = ORT_ERR_HIP; return;
                }
                if (rc == ORT_OK) rc = ort_render_image_device(s, &q, packed[(size_t)r], nullptr, &stats[(size_t)r]);
                if (rc != ORT_OK) errs[(size_t)r] = ort_last_error();
                rcs[(size_t)r] = rc;
            });
        }
        for (auto &w : workers) w.join();
        for (int r = 0; r < gpus; ++r)
            if (rcs[(size_t)r] != ORT_OK) { fprintf(stderr, "GPU %d: %s\n", r, errs[(size_t)r].c_str()); return 1; }
        std::vector<ort_comm *> comms((size_t)gpus, nullptr);
        void *full = nullptr;
        if (ort_comm_create_local(gpus, devices.data(), comms.data()) != ORT_OK || hipSetDevice(0) != hipSuccess ||
            hipMalloc(&full, image.size() * 4u) != hipSuccess ||
            ort_gather_framebuffer_local(comms.data(), gpus, (const void *const *)packed.data(), full, width, height, nullptr) != ORT_OK ||
            hipDeviceSynchronize() != hipSuccess || hipMemcpy(image.data(), full, image.size() * 4u, hipMemcpyDeviceToHost) != hipSuccess) {
            fprintf(stderr, "framebuffer gather failed: %s\n", ort_last_error());
            return 1;
        }
        for (int r = 0; r < gpus; ++r) {
            st.kernel_ms = stats[(size_t)r].kernel_ms > st.kernel_ms ? stats[(size_t)r].kernel_ms : st.kernel_ms;
            ort_comm_destroy(comms[(size_t)r]);
            (void)hipSetDevice(devices[(size_t)r]);
            (void)hipFree(packed[(size_t)r]);
            if (r != 0) ort_scene_destroy(scenes[(size_t)r]);
        }
        (void)hipSetDevice(0);
        (void)hipFree(full);
    } else if (ort_render_image(scene, &p, image.data(), &st) != ORT_OK) {
        fprintf(stderr, "render failed: %s\n", ort_last_error());
        return 1;
    }
    double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    double paths = (double)width * height * spp;
    printf("rendered %dx%d, %u spp, %u triangles on %d GPU(s): %.3f s wall, %.3f ms kernel%s, %.2f Mpaths/s (kernel)\n", width, height, spp,
           info.triangle_count, gpus > 1 ? gpus : 1, sec, st.kernel_ms, gpus > 1 ? " (slowest GPU)" : "", paths / (st.kernel_ms * 1e-3) * 1e-6);
    if (!raw_path.empty()) {
        FILE *f = fopen(raw_path.c_str(), "wb");
        if (f) { fwrite(image.data(), 4, image.size(), f); fclose(f); }
    }
    if (ort_write_hdr(out_path.c_str(), image.data(), width, height) != ORT_OK) {
        fprintf(stderr, "cannot write %s\n", out_path.c_str());
        return 1;
    }
    ort_scene_destroy(scene);
    return 0;
}
