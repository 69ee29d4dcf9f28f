"""BASELINE-size checks through size-independent properties (the oracle cannot render 2e9 paths):
determinism, shard union == whole, and oracle agreement on sub-rects of the full-size frame."""
import numpy as np
import pytest

from conftest import assert_bits_equal

pytestmark = pytest.mark.gpu

W, H = 1920, 1080


def test_full_frame_subrects_match_oracle(api, oracle, gpu_scene):
    """full 1080p frame at 16 spp (CHUNK 8): three 24x16 windows (bunny, wall, ceiling light)
    must equal the oracle rendering only those windows of the same 1920x1080 image."""
    scene = gpu_scene("c3_bunny_room")
    spp, chunk, seed = 16, 8, 12345
    img, st = scene.render(W, H, spp, seed, "chunk", chunk=chunk, counters=True)
    assert st["paths"] == W * H * spp
    osc = oracle.OracleScene(scene.flatten(W, H))
    for (x0, y0) in [(930, 500), (100, 900), (1500, 60)]:
        rect = (x0, y0, x0 + 24, y0 + 16)
        ref, _ = osc.render(W, H, spp, seed, "chunk", chunk=chunk, rect=rect, threads=16)
        assert_bits_equal(img[rect[1]:rect[3], rect[0]:rect[2]], ref[rect[1]:rect[3], rect[0]:rect[2]], "window %s" % (rect,))


def test_full_frame_determinism_and_shards(api, gpu_scene):
    scene = gpu_scene("c3_bunny_room")
    spp, chunk, seed = 8, 8, 99
    a, _ = scene.render(W, H, spp, seed, "chunk", chunk=chunk)
    b, _ = scene.render(W, H, spp, seed, "chunk", chunk=chunk)
    assert_bits_equal(a, b, "two runs")
    acc = np.zeros_like(a)
    for r in range(8):
        part, _ = scene.render(W, H, spp, seed, "chunk", chunk=chunk, shard=(r, 8))
        acc += part
    assert_bits_equal(acc, a, "8-way shard union")
    assert np.isfinite(a).all()


def test_full_frame_ray_exchange(api, gpu_scene, monkeypatch):
    """the whole 1920x1080 frame (2 M pixels, 32 spp in 8-sample jobs = 32 jobs per lane): exchange forced off == forced on
    == left to itself (on from 24 jobs per lane), bit for bit"""
    scene = gpu_scene("c3_bunny_room")
    spp, chunk, seed = 32, 8, 2025
    monkeypatch.setenv("ORT_EXCHANGE", "0")
    a, _ = scene.render(W, H, spp, seed, "chunk", chunk=chunk)
    monkeypatch.setenv("ORT_EXCHANGE", "1")
    b, _ = scene.render(W, H, spp, seed, "chunk", chunk=chunk)
    assert_bits_equal(a, b, "exchange on vs off, full frame")
    monkeypatch.delenv("ORT_EXCHANGE")
    c, _ = scene.render(W, H, spp, seed, "chunk", chunk=chunk)
    assert_bits_equal(a, c, "default launch policy, full frame")


def test_headline_config_window_matches_oracle(api, oracle, gpu_scene):
    """BASELINE.json's headline parameters themselves -- 1920x1080, 1024 spp as 16 serial 64-sample jobs per pixel --
    on a 24x16 window over the bunny (393 216 paths: ~1 s of oracle time): bit-equal to the oracle, and chunk 0 is the
    PIXEL policy's 64-spp image (the seeds line up)."""
    scene = gpu_scene("c3_bunny_room")
    rect = (940, 520, 964, 536)
    seed = 12345
    full, _ = scene.render(W, H, 1024, seed, "chunk", chunk=64, rect=rect)
    ref, _ = oracle.OracleScene(scene.flatten(W, H)).render(W, H, 1024, seed, "chunk", chunk=64, rect=rect, threads=16)
    assert_bits_equal(full[rect[1]:rect[3], rect[0]:rect[2]], ref[rect[1]:rect[3], rect[0]:rect[2]], "1024 spp / chunk 64 window")
    k0, _ = scene.render(W, H, 64, seed, "chunk", chunk=64, rect=rect)  # == chunk 0 of the 1024-spp job
    px, _ = scene.render(W, H, 64, seed, "pixel", rect=rect)
    assert_bits_equal(k0, px, "chunk 0 == pixel policy")


# ---- BASELINE.json configs[3] and [4]: 3840x2160 -------------------------------------------------
W4, H4 = 3840, 2160


def test_c4_dwarf_4k_windows_match_oracle(api, oracle, gpu_scene):
    """dwarf.obj room at 3840x2160 (the 8-GPU config): windows of the full frame against the oracle,
    and the 8-way shard union of the frame equals the single-GPU frame"""
    scene = gpu_scene("c4_dwarf_room")
    spp, chunk, seed = 8, 4, 12345
    img, st = scene.render(W4, H4, spp, seed, "chunk", chunk=chunk, counters=True)
    assert st["paths"] == W4 * H4 * spp
    osc = oracle.OracleScene(scene.flatten(W4, H4))
    for (x0, y0) in [(1900, 1000), (200, 1800), (3000, 120)]:
        rect = (x0, y0, x0 + 24, y0 + 16)
        ref, _ = osc.render(W4, H4, spp, seed, "chunk", chunk=chunk, rect=rect, threads=16)
        assert_bits_equal(img[rect[1]:rect[3], rect[0]:rect[2]], ref[rect[1]:rect[3], rect[0]:rect[2]], "window %s" % (rect,))
    acc = np.zeros_like(img)
    for r in range(8):
        part, _ = scene.render(W4, H4, spp, seed, "chunk", chunk=chunk, shard=(r, 8))
        acc += part
    assert_bits_equal(acc, img, "8-way shard union")


def test_c5_million_triangle_mesh(api, oracle, gpu_scene):
    """the 999 698-triangle height field (deep-tree stress config) at 3840x2160.  The reference build
    cannot hold this mesh (its fixed arenas overflow), so the full size is pinned by the oracle
    -- itself pinned by the reference on the 99 458-triangle decimation (golden fixtures) -- on windows
    of the frame, plus determinism and shard composition."""
    scene = gpu_scene("c5_heightfield_708")
    assert scene.info().triangle_count == 999698
    spp, chunk, seed = 4, 2, 12345
    img, st = scene.render(W4, H4, spp, seed, "chunk", chunk=chunk, counters=True)
    assert st["paths"] == W4 * H4 * spp
    assert np.isfinite(img).all()
    again, _ = scene.render(W4, H4, spp, seed, "chunk", chunk=chunk)
    assert_bits_equal(img, again, "two runs")
    osc = oracle.OracleScene(scene.flatten(W4, H4))
    for (x0, y0) in [(1900, 1000), (1500, 1200), (2400, 900), (300, 300)]:
        rect = (x0, y0, x0 + 16, y0 + 8)
        ref, _ = osc.render(W4, H4, spp, seed, "chunk", chunk=chunk, rect=rect, threads=16)
        assert_bits_equal(img[rect[1]:rect[3], rect[0]:rect[2]], ref[rect[1]:rect[3], rect[0]:rect[2]], "window %s" % (rect,))
    acc = np.zeros_like(img)
    for r in range(2):
        part, _ = scene.render(W4, H4, spp, seed, "chunk", chunk=chunk, shard=(r, 2))
        acc += part
    assert_bits_equal(acc, img, "2-way shard union")


# ---- every BASELINE.json config at ITS OWN parameters (spp, 64-sample jobs), windows against the oracle ------------
# Long serial jobs are where one divergent comparison costs a whole job, so the short-job tests above do not cover them.
def _windows_vs_oracle(scene, oracle, w, h, spp, chunk, seed, windows, what):
    osc = oracle.OracleScene(scene.flatten(w, h))
    for (x0, y0, ww, hh) in windows:
        rect = (x0, y0, x0 + ww, y0 + hh)
        img, _ = scene.render(w, h, spp, seed, "chunk", chunk=chunk, rect=rect)
        ref, _ = osc.render(w, h, spp, seed, "chunk", chunk=chunk, rect=rect, threads=16)
        assert_bits_equal(img[rect[1]:rect[3], rect[0]:rect[2]], ref[rect[1]:rect[3], rect[0]:rect[2]], "%s window %s" % (what, rect))


def test_c2_headline_parameters_windows_match_oracle(api, oracle, gpu_scene):
    """BASELINE.json configs[1]: the analytic scene at 1920x1080, 1024 spp in 64-sample jobs, all-lobes kernel flavour --
    windows over the glass sphere, the mirror box, the red glass sphere, a mirror sphere of the table frame and the
    second glossy box (positions from the camera projection of the shapes' centres)"""
    scene = gpu_scene("c2_analytic")
    _windows_vs_oracle(scene, oracle, W, H, 1024, 64, 12345,
                       [(1026, 714, 24, 16), (708, 176, 24, 16), (1233, 495, 24, 16), (1281, 252, 16, 12), (683, 549, 24, 16)], "c2 1080p/1024/64")


def test_c4_headline_parameters_windows_match_oracle(api, oracle, gpu_scene):
    """BASELINE.json configs[3]: dwarf room at 3840x2160, 512 spp in 64-sample jobs: two windows on the dwarf, one on the floor"""
    scene = gpu_scene("c4_dwarf_room")
    _windows_vs_oracle(scene, oracle, W4, H4, 512, 64, 12345, [(1835, 1292, 24, 16), (1888, 992, 24, 16), (600, 400, 24, 16)], "c4 4K/512/64")


def test_c5_full_mesh_64_sample_jobs_window_matches_oracle(api, oracle, gpu_scene):
    """BASELINE.json configs[4]'s job shape on the FULL 999 698-triangle mesh: 3840x2160, one 64-sample job per pixel, a 16x8
    window over the height field (the oracle walks the reference's own octree: ~5000 triangle tests per ray)"""
    scene = gpu_scene("c5_heightfield_708")
    _windows_vs_oracle(scene, oracle, W4, H4, 64, 64, 12345, [(1900, 1000, 16, 8)], "c5 4K/64/64")


def test_hip_image_against_the_reference_at_baseline_scale(api, gpu_scene, manifest):
    """north_star: "per-pixel L2 error < 1e-4 vs CPU reference", asserted on GPU OUTPUT, not by transitivity.  Fixtures
    (tests/golden/baseline_windows.npz, tools/glibc_distance_baseline.py): the 128x72 central window of the 1920x1080 frame
    at 1024 spp / 64-sample jobs rendered by the reference's own sources compiled in the dev container, once with the
    deterministic libm (the parity anchor: the HIP image must equal it BIT FOR BIT) and once with glibc's libm (the
    reference as it ships on this platform: the HIP image must be within the stated tolerance).  Stated bound: c3 and c4
    >= 99 % of the pixels bit-equal and EVERY pixel within 1e-4 (observed: 1.5e-8); c2 (glass and mirrors: one flipped
    comparison decorrelates a whole 64-sample job) >= 99 % bit-equal, >= 99.9 % within 1e-4, at most 2 pixels beyond, none
    beyond 1e-2."""
    import os
    from conftest import GOLDEN
    z = np.load(os.path.join(GOLDEN, "baseline_windows.npz"))
    for name in ("c3_bunny_room", "c4_dwarf_room", "c2_analytic"):
        meta = manifest["glibc_distance_baseline"][name]
        x0, y0, x1, y1 = meta["window"]
        assert (meta["width"], meta["height"], meta["spp"], meta["chunk"]) == (W, H, 1024, 64)
        scene = gpu_scene(name)
        img, _ = scene.render(W, H, 1024, meta["seed"], "chunk", chunk=64, rect=(x0, y0, x1, y1))
        got = img[y0:y1, x0:x1]
        assert_bits_equal(got, z[name + "__det"], "%s vs reference + deterministic libm" % name)
        g = z[name + "__glibc"]
        l2 = np.sqrt(((got.astype(np.float64) - g.astype(np.float64)) ** 2).sum(axis=2)).ravel()
        bit_equal = float((got.view("<u4") == g.view("<u4")).all(axis=2).mean())
        assert bit_equal >= 0.99, (name, bit_equal)
        if name == "c2_analytic":
            assert (l2 < 1e-4).mean() >= 0.999 and int((l2 >= 1e-4).sum()) <= 2 and l2.max() < 1e-2, (name, l2.max(), int((l2 >= 1e-4).sum()))
        else:
            assert l2.max() < 1e-4, (name, l2.max())
