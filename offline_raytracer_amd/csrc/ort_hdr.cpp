/*
 * ort_hdr.cpp -- Radiance .hdr output, byte-compatible with the reference writer
 * (code/macos_main.mm:242-261 v3_to_rgbe, :263-287 header, :683-707 row loop):
 * header "#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n+Y H +X W\n", then uncompressed RGBE
 * pixels (despite the "rle" in the header), buffer row H-1 first down to row 0.
 */
#include <math.h>
#include <stdio.h>

#include <vector>

#include "ort_scene.h"

namespace ort {

uint32_t rgbe_pack(float r, float g, float b) {
    float m = (r > g) ? r : g; /* maximum(maximum(r,g),b), types.h:50 */
    m = (m > b) ? m : b;
    uint32_t out = 0;
    if (m >= 1e-32f) {
        int e;
        float denom = frexpf(m, &e) * 255.0f / m; /* 255, not 256 */
        out = ((uint32_t)roundf(r * denom) << 0) | ((uint32_t)roundf(g * denom) << 8) |
              ((uint32_t)roundf(b * denom) << 16) | ((uint32_t)(e + 128) << 24);
    }
    return out;
}

} // namespace ort

extern "C" uint32_t ort_rgbe(float r, float g, float b) { return ort::rgbe_pack(r, g, b); }

extern "C" int ort_write_hdr(const char *path, const float *rgb, int32_t width, int32_t height) {
    if (!path || !rgb || width <= 0 || height <= 0) return ORT_ERR_INVALID;
    FILE *f = fopen(path, "wb");
    if (!f) return ORT_ERR_IO;
    fprintf(f, "#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n+Y %d +X %d\n", height, width);
    std::vector<uint32_t> row((size_t)width);
    for (int32_t y = height - 1; y >= 0; --y) {
        const float *p = rgb + 3 * (size_t)y * (size_t)width;
        for (int32_t x = 0; x < width; ++x) row[(size_t)x] = ort::rgbe_pack(p[3 * x], p[3 * x + 1], p[3 * x + 2]);
        if (fwrite(row.data(), 4, (size_t)width, f) != (size_t)width) { fclose(f); return ORT_ERR_IO; }
    }
    return fclose(f) == 0 ? ORT_OK : ORT_ERR_IO;
}
