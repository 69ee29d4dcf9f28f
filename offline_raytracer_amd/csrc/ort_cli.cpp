/*
 * ort_render -- command-line driver: the Linux/MI355X replacement for the reference's
 * main() (code/macos_main.mm:289-710), with its literals turned into flags: scene path
 * (:317), resolution (:319-320), samples per pixel (:612), seed (:297-298), roulette (:656).
 * Loads a .scn, builds the tree, uploads to a GPU, renders through the C ABI, writes .hdr.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <chrono>
#include <string>
#include <vector>

#include "../../include/ort.h"

static void usage(const char *argv0) {
    fprintf(stderr,
            "usage: %s --scene file.scn [--base dir/] [--width W --height H] [--spp N] [--seed S]\n"
            "          [--policy tile32|whole|pixel|chunk] [--chunk C] [--rr 0.8] [--device D]\n"
            "          [--out image.hdr] [--raw image.f32]\n",
            argv0);
}

int main(int argc, char **argv) {
    std::string scene_path, base, out_path = "output.hdr", raw_path, policy = "chunk";
    int width = 0, height = 0, device = 0;
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
    if (ort_scene_commit(scene) != ORT_OK || ort_scene_upload(scene, device) != ORT_OK) {
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
    if (ort_render_image(scene, &p, image.data(), &st) != ORT_OK) {
        fprintf(stderr, "render failed: %s\n", ort_last_error());
        return 1;
    }
    double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    double paths = (double)width * height * spp;
    printf("rendered %dx%d, %u spp, %u triangles: %.3f s wall, %.3f ms kernel, %.2f Mpaths/s (kernel)\n", width, height, spp,
           info.triangle_count, sec, st.kernel_ms, paths / (st.kernel_ms * 1e-3) * 1e-6);
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
