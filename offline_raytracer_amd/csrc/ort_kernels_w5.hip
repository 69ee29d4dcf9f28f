/*
 * ort_kernels_w5.hip -- the plain-loop path-trace kernels at FIVE waves per SIMD.
 *
 * The lane code (ort_lane.h) compiled a second time, with other limits: 96 vector registers per lane instead of 128
 * (5 waves per SIMD instead of 4), 20 LDS traversal-stack entries per lane instead of 24 (five workgroups' LDS must fit one CU:
 * 5 x 30.4 KB), and -- the Makefile's doing -- with LLVM's machine LICM switched off: hoisted out of the one big lane loop, the
 * binary64 constants of the libm polynomials and the stash / table addresses are held in registers across everything and
 * spilled (all-lobes flavour at 96 registers: 90 spilled vector registers and 206 spilled scalar ones with it, 60 and 49 without).
 * Why (profiles/r03_tuning.md): the kernel is bound by memory latency and instruction issue, and a fifth wave per SIMD hides more
 * of both than the spills cost -- for the all-lobes flavour (analytic scene 3 302 -> 3 514 Mpaths/s, glass room 3 625 -> 3 848) and
 * for trees that leave the L2 (1M-triangle scene 1 401 -> 1 500); three waves lose 13-15 %, six lose again (161-190 spilled
 * registers).  The diffuse flavour on cache-resident trees gains nothing (and its small shards lose: more lanes, fewer jobs per
 * lane), the ray exchange loses 2 %: those stay at four waves, in ort_kernels.hip proper.  device_render chooses; ORT_WAVES5=0 / 1
 * forces.  Same lane code, same results (tests/test_gpu_parity.py::test_five_waves_build_equals_four_waves_build).
 */
#define ORT_W5_TU 1
#define ORT_WAVES_PER_EU 5
#define ORT_LDS_STACK 20
#define ORT_SPILL_STACK 44
#include "ort_lane.h"

/* the five-waves variants of the plain loop, launched by device_render (ort_kernels.hip proper).  The argument structs are
   the same declarations compiled in this unit's namespace: passed as bytes */
void ort_launch_w5(int diffuse, unsigned int grid, void *stream, const void *sv_bytes, const void *hot_bytes) {
    ort_w5::SceneView sv;
    ort_w5::RenderHot hot;
    memcpy(&sv, sv_bytes, sizeof(sv));
    memcpy(&hot, hot_bytes, sizeof(hot));
    if (diffuse) hipLaunchKernelGGL((ort_w5::pt_persistent<false, true, true, true>), dim3(grid), dim3(ort_w5::kBlock), 0, (hipStream_t)stream, sv, hot);
    else hipLaunchKernelGGL((ort_w5::pt_persistent<false, false, true, true>), dim3(grid), dim3(ort_w5::kBlock), 0, (hipStream_t)stream, sv, hot);
}
size_t ort_w5_sizeof_scene_view() { return sizeof(ort_w5::SceneView); }
size_t ort_w5_sizeof_render_hot() { return sizeof(ort_w5::RenderHot); }
