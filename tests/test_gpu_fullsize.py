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
    """the whole 1920x1080 frame (2 M pixels, 32 spp in 8-sample jobs = 32 jobs per lane: the launch shape on which the
    exchange switches itself on): exchange forced off == forced on == left to itself, bit for bit"""
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
