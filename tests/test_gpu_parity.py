"""Parity tests proper: the HIP path (through the C ABI) against the oracle on the same seeded
inputs, and against the committed golden vectors made by the reference's own code.  Everything
here is BIT-EXACT: the path computes in f32 with separately rounded operations on both sides
(-ffp-contract=off), IEEE divide/sqrt, and a shared deterministic libm."""
import os

import numpy as np
import pytest

import ref_io
from conftest import DATA, GOLDEN, ROOT, assert_bits_equal

pytestmark = pytest.mark.gpu

SCENES = ["testscene", "c2_analytic", "c3_bunny_room", "c4_dwarf_room", "letters", "glass_room", "rand_a", "rand_b"]
# + the 99 458-triangle decimation of BASELINE.json's synthetic-mesh config (generated, see conftest.load_scene)
SCENES_C5 = SCENES + ["c5_heightfield_224"]


def test_device_present(api):
    assert api.device_count() >= 1


# ---- per-function parity on the device ------------------------------------------------------
def test_unit_tables_on_device(api):
    """intersectors, BSDF sample/pdf/eval, lobe sampling, libm, normalize: the device functions
    against the reference's outputs (ops 1-4: t and normal; inner_hit is not produced)."""
    z = np.load(os.path.join(GOLDEN, "unit_tables.npz"))
    recs = z["records"].view(ref_io.UNIT_REC_DTYPE).reshape(-1)
    got = api.unit_eval_device(recs)
    want = z["ref_det"]
    for op in np.unique(recs["op"]):
        sel = recs["op"] == op
        cols = 4 if op <= 4 else 8
        assert_bits_equal(got[sel][:, :cols], want[sel][:, :cols], "device unit op %d" % op)


def test_ieee_arithmetic_and_rng_on_device(api, oracle):
    """the premise of bit-exactness: f32 / sqrt * + - and u32->f32 on gfx950 equal x86 SSE,
    denormals included; xorshift and the job seed hash equal the oracle's."""
    rng = np.random.default_rng(11)
    n = 20000
    a = rng.integers(0, 2 ** 32, size=(n, 3), dtype=np.uint64).astype("<u4").view("<f4")
    a[:2000] = (10.0 ** rng.uniform(-44, -36, size=(2000, 3))).astype("<f4")  # denormal range
    a[2000:4000, 1] = a[2000:4000, 0] * np.float32(3.0)
    recs = ref_io.make_unit_records(13, a)
    got = api.unit_eval_device(recs)
    with np.errstate(all="ignore"):
        x, y, c = a[:, 0], a[:, 1], a[:, 2]
        want = np.stack([x / y, np.sqrt(x), x * y, x + y, x - y, x.view("<u4").astype("<f4"), (x * y).astype("<f4") + c], axis=1)
    ok = (got[:, :7].view("<u4") == want.astype("<f4").view("<u4")) | (np.isnan(got[:, :7]) & np.isnan(want))
    assert ok.all(), "IEEE mismatch in columns %s" % np.unique(np.argwhere(~ok)[:, 1])
    seeds = rng.integers(1, 2 ** 32, size=(4096, 2), dtype=np.uint64).astype("<u4")
    got = api.unit_eval_device(ref_io.make_unit_records(12, seeds.view("<f4")))
    for i in range(0, 4096, 97):
        tab = np.frombuffer(oracle.rng_table(int(seeds[i, 0]), 2), dtype="<u4")
        assert got[i, 0].view("<u4") == tab[0]
        assert got[i, 1].view("<u4") == tab[1]  # rng_01 of the first step
        assert got[i, 4].view("<u4") == oracle.job_seed(int(seeds[i, 0]), int(seeds[i, 1]))


# ---- renders ---------------------------------------------------------------------------------
@pytest.mark.parametrize("name", SCENES_C5)
def test_renders_match_reference_goldens(api, manifest, gpu_scene, name):
    """GPU image == the reference's own image (golden), every seeding policy."""
    z = np.load(os.path.join(GOLDEN, "renders_%s.npz" % name))
    scene = gpu_scene(name)
    for e in [e for e in manifest["renders"] if e["scene"] == name]:
        policy = "chunk" if e["policy"] == "sample" else e["policy"]
        img, _ = scene.render(e["width"], e["height"], e["spp"], e["seed"], policy, chunk=e["chunk"])
        assert_bits_equal(img, z[e["key"]], "%s %s" % (name, e["key"]))


@pytest.mark.parametrize("name", SCENES_C5)
@pytest.mark.parametrize("policy,w,h,spp,chunk", [("chunk", 96, 64, 16, 4), ("pixel", 80, 50, 6, 0), ("chunk", 37, 23, 5, 1)])
def test_renders_match_oracle(api, oracle, gpu_scene, name, policy, w, h, spp, chunk):
    scene = gpu_scene(name)
    img, st = scene.render(w, h, spp, 4242, policy, chunk=chunk, counters=True)
    ref, ost = oracle.OracleScene(scene.flatten(w, h)).render(w, h, spp, 4242, policy, chunk=max(chunk, 1), threads=16)
    assert_bits_equal(img, ref, "%s %s" % (name, policy))
    assert st["paths"] == w * h * spp == ost["paths"]
    assert st["rays"] == ost["rays"]  # same paths => same number of rays cast


def test_single_call_is_the_reference_call(api, oracle, gpu_scene):
    """ort_tiled_raytrace == tiled_raytrace_bvh (ray.cpp:1178): same rect, same RandomSeries in,
    same pixels and same RandomSeries out; pixels outside the rect are untouched."""
    scene = gpu_scene("c4_dwarf_room")
    w, h = 32, 24
    out = np.full((h, w, 3), 7.0, "<f4")
    ref = np.full((h, w, 3), 7.0, "<f4")
    osc = oracle.OracleScene(scene.flatten(w, h))
    state = 123456789
    for rect in [(3, 2, 9, 5), (10, 10, 11, 11), (0, 20, 32, 24)]:
        _, s_gpu = scene.tiled_raytrace(out, *rect, state, 3)
        _, s_ref = osc.tiled_raytrace(ref, *rect, state, 3)
        assert s_gpu == s_ref
        state = s_gpu
    assert_bits_equal(out, ref)
    assert (out[0, 0] == 7.0).all()


def test_batch_jobs_and_empty_rects(api, oracle, gpu_scene):
    scene = gpu_scene("glass_room")
    w, h = 40, 30
    jobs = np.zeros(5, api.JOB_DTYPE)
    jobs[0] = (0, 0, 8, 8, 11, 2)
    jobs[1] = (8, 0, 40, 3, 12, 1)
    jobs[2] = (5, 10, 5, 20, 13, 4)      # empty rect: zero iterations, state unchanged
    jobs[3] = (39, 29, 40, 30, 0xFFFFFFFF, 7)
    jobs[4] = (0, 29, 39, 30, 1, 3)
    out = np.zeros((h, w, 3), "<f4")
    finals, _ = scene.tiled_raytrace_batch(out, jobs)
    ref = np.zeros((h, w, 3), "<f4")
    osc = oracle.OracleScene(scene.flatten(w, h))
    for i, j in enumerate(jobs):
        _, s = osc.tiled_raytrace(ref, int(j["x0"]), int(j["y0"]), int(j["x1"]), int(j["y1"]), int(j["rng_state"]), int(j["spp"]))
        assert finals[i] == s, i
    assert finals[2] == 13
    assert_bits_equal(out, ref)


def test_rect_and_shards_compose(api, gpu_scene):
    """a sub-rect equals the same pixels of the full render; the union of N shards equals the
    unsharded render (what makes the multi-GPU path correct by construction)."""
    scene = gpu_scene("c3_bunny_room")
    w, h, spp = 70, 45, 4
    full, _ = scene.render(w, h, spp, 9, "chunk", chunk=2)
    part, _ = scene.render(w, h, spp, 9, "chunk", chunk=2, rect=(13, 7, 50, 31))
    assert_bits_equal(part[7:31, 13:50], full[7:31, 13:50])
    assert (part[:7] == 0).all() and (part[:, :13] == 0).all()
    for world in (2, 3, 8):
        acc = np.zeros_like(full)
        for r in range(world):
            img, _ = scene.render(w, h, spp, 9, "chunk", chunk=2, shard=(r, world))
            touched = (img != 0).any(axis=2)
            assert not (touched & (acc != 0).any(axis=2)).any(), "shards overlap"
            acc += img
        assert_bits_equal(acc, full, "world %d" % world)


def test_pixel_policy_is_chunk_with_one_chunk(api, gpu_scene):
    scene = gpu_scene("c2_analytic")
    a, _ = scene.render(48, 32, 6, 5, "pixel")
    b, _ = scene.render(48, 32, 6, 5, "chunk", chunk=6)
    assert_bits_equal(a, b)


def test_wavefront_mode_equals_persistent_mode(api, gpu_scene, monkeypatch):
    """the two execution modes (one persistent kernel / HBM-resident wavefront) run the same lane code"""
    scene = gpu_scene("c4_dwarf_room")
    monkeypatch.setenv("ORT_MODE", "persistent")
    a, _ = scene.render(100, 60, 8, 3, "chunk", chunk=4)
    monkeypatch.setenv("ORT_MODE", "wavefront")
    b, st = scene.render(100, 60, 8, 3, "chunk", chunk=4, counters=True)
    assert st["paths"] == 100 * 60 * 8
    assert_bits_equal(a, b)
    out = np.zeros((24, 32, 3), "<f4")
    ref = np.zeros((24, 32, 3), "<f4")
    jobs = np.zeros(3, api.JOB_DTYPE)
    jobs[0] = (0, 0, 8, 8, 11, 2); jobs[1] = (8, 0, 32, 3, 12, 1); jobs[2] = (31, 23, 32, 24, 5, 7)
    fw, _ = scene.tiled_raytrace_batch(out, jobs)
    monkeypatch.setenv("ORT_MODE", "persistent")
    fp, _ = scene.tiled_raytrace_batch(ref, jobs)
    assert (fw == fp).all()
    assert_bits_equal(out, ref)


@pytest.mark.parametrize("name", ["c3_bunny_room", "c4_dwarf_room"])
def test_diffuse_kernel_equals_general_kernel(api, gpu_scene, monkeypatch, name):
    """scenes whose materials all have Ks = Kt = 0 run the kernel compiled without the specular /
    transmission evaluation; ORT_KERNEL=general forces the all-lobes kernel: same bits"""
    scene = gpu_scene(name)
    monkeypatch.delenv("ORT_KERNEL", raising=False)
    a, _ = scene.render(160, 90, 32, 21, "chunk", chunk=8)
    monkeypatch.setenv("ORT_KERNEL", "general")
    b, _ = scene.render(160, 90, 32, 21, "chunk", chunk=8)
    assert_bits_equal(a, b)


@pytest.mark.parametrize("name,policy,w,h,spp,chunk", [("c3_bunny_room", "chunk", 200, 120, 32, 8), ("c4_dwarf_room", "pixel", 160, 90, 24, 0),
                                                     ("testscene", "chunk", 128, 72, 16, 4), ("glass_room", "chunk", 128, 72, 16, 8)])
def test_ray_exchange_equals_plain_loop(api, oracle, gpu_scene, monkeypatch, name, policy, w, h, spp, chunk):
    """the ray exchange (stragglers park in the wave's stash in HBM, the shading pass runs full, parked rays are traversed
    64 at a time; on by itself only for long diffuse launches) only changes WHICH lane advances a path and WHEN: same
    bits as the plain loop and as the oracle, in both kernel flavours, with the long-phase threshold at its default
    and squeezed so that every mechanism (parking with deep stacks, refills, the drain at the end) is exercised"""
    scene = gpu_scene(name)
    monkeypatch.setenv("ORT_EXCHANGE", "0")
    a, _ = scene.render(w, h, spp, 17, policy, chunk=chunk)
    monkeypatch.setenv("ORT_EXCHANGE", "1")
    b, _ = scene.render(w, h, spp, 17, policy, chunk=chunk)
    assert_bits_equal(a, b, "exchange on vs off")
    # the last two sets are the ones that could leave a wave with every lane holding off new jobs and too few parked rays to
    # start a traversal phase (inflight cap below the long-phase threshold): the host clamps the cap (device_render)
    for knobs in ({"ORT_LONG_MIN": "16", "ORT_INFLIGHT_CAP": "16", "ORT_REFILL_BELOW": "40"}, {"ORT_LONG_MIN": "128", "ORT_INFLIGHT_CAP": "128", "ORT_LONG_REFILL": "60"},
                  {"ORT_INFLIGHT_CAP": "0"}, {"ORT_LONG_MIN": "128"}):
        for k, v in knobs.items():
            monkeypatch.setenv(k, v)
        c, _ = scene.render(w, h, spp, 17, policy, chunk=chunk)
        assert_bits_equal(a, c, "exchange with %s" % knobs)
        for k in knobs:
            monkeypatch.delenv(k)
    ref, _ = oracle.OracleScene(scene.flatten(w, h)).render(w, h, spp, 17, policy, chunk=max(chunk, 1), threads=16)
    assert_bits_equal(b, ref, "exchange vs oracle")


@pytest.mark.parametrize("name,w,h,spp,chunk", [("c3_bunny_room", 256, 144, 16, 4), ("testscene", 200, 120, 8, 4), ("c2_analytic", 160, 90, 8, 2),
                                                  ("c5_heightfield_224", 256, 144, 8, 4), ("rand_b", 160, 90, 8, 8)])
def test_wide_tree_equals_binary_tree(api, oracle, gpu_scene, monkeypatch, name, w, h, spp, chunk):
    """the 4-wide form of the fast tree (128-byte nodes, visit_node4: what trees that leave the L2 are traversed with) only
    changes the order in which a ray meets the boxes: same bits as the binary tree and as the oracle, production and
    counters kernels, also with every 16th ray re-cast exactly"""
    scene = gpu_scene(name)
    assert scene.tree_info()["wide_node_count"] > 0
    monkeypatch.setenv("ORT_WIDE", "0")
    a, sa = scene.render(w, h, spp, 23, "chunk", chunk=chunk, counters=True)
    monkeypatch.setenv("ORT_WIDE", "1")
    b, sb = scene.render(w, h, spp, 23, "chunk", chunk=chunk, counters=True)
    c, _ = scene.render(w, h, spp, 23, "chunk", chunk=chunk)
    assert_bits_equal(a, b, "wide vs binary (counters kernels)")
    assert_bits_equal(a, c, "wide production kernel")
    assert sa["rays"] == sb["rays"] and sa["paths"] == sb["paths"]
    if name in ("c3_bunny_room", "c5_heightfield_224", "testscene"):
        assert sb["node_tests"] != sa["node_tests"]  # it really was another tree
    monkeypatch.setenv("ORT_DEBUG_FORCE_FALLBACK", "0xf")
    e, _ = scene.render(w, h, spp, 23, "chunk", chunk=chunk)
    assert_bits_equal(a, e, "wide tree with forced exact re-casts")
    monkeypatch.delenv("ORT_DEBUG_FORCE_FALLBACK")
    ref, _ = oracle.OracleScene(scene.flatten(w, h)).render(w, h, spp, 23, "chunk", chunk=chunk, threads=16)
    assert_bits_equal(b, ref, "wide tree vs oracle")


@pytest.mark.parametrize("name,w,h,spp,chunk", [("c2_analytic", 200, 120, 16, 4), ("c3_bunny_room", 256, 144, 16, 4), ("c5_heightfield_224", 256, 144, 8, 4),
                                                  ("glass_room", 160, 90, 16, 4)])
def test_five_waves_build_equals_four_waves_build(api, oracle, gpu_scene, monkeypatch, name, w, h, spp, chunk):
    """the plain loop exists twice: at four waves per SIMD (128 registers, 24 LDS stack entries) and at five (96 registers, 20
    entries, ort_kernels_w5.hip -- chosen by itself for the all-lobes flavour and for trees that leave the L2).  Same lane code:
    same bits, CHUNK and PIXEL policies, also with every 16th ray re-cast exactly (resolve_hit's re-traversals use the shorter
    LDS stack), and equal to the oracle"""
    scene = gpu_scene(name)
    monkeypatch.setenv("ORT_EXCHANGE", "0")
    monkeypatch.setenv("ORT_WAVES5", "0")
    a, _ = scene.render(w, h, spp, 29, "chunk", chunk=chunk)
    ap, _ = scene.render(w, h, spp, 29, "pixel")
    monkeypatch.setenv("ORT_WAVES5", "1")
    b, _ = scene.render(w, h, spp, 29, "chunk", chunk=chunk)
    bp, _ = scene.render(w, h, spp, 29, "pixel")
    assert_bits_equal(a, b, "five waves vs four, CHUNK")
    assert_bits_equal(ap, bp, "five waves vs four, PIXEL")
    monkeypatch.setenv("ORT_DEBUG_FORCE_FALLBACK", "0xf")
    c, _ = scene.render(w, h, spp, 29, "chunk", chunk=chunk)
    assert_bits_equal(a, c, "five waves with forced exact re-casts")
    monkeypatch.delenv("ORT_DEBUG_FORCE_FALLBACK")
    ref, _ = oracle.OracleScene(scene.flatten(w, h)).render(w, h, spp, 29, "chunk", chunk=chunk, threads=16)
    assert_bits_equal(b, ref, "five waves vs oracle")


def test_issue_order_of_chunk_jobs_does_not_matter(api, gpu_scene, monkeypatch):
    """CHUNK renders issue their jobs block-major -- [block][chunk][pixel]: all chunks of an 8x8 block together
    (ort_lane.h, "the order in which a CHUNK render issues its jobs") -- instead of chunk-major as in rounds 1-2
    (ORT_LPT=0).  Seeds belong to jobs, so the image is the same bit for bit: whole frames with ragged edge blocks, clipped
    rects, 3-way shards, plain loop and ray exchange (whose waves stop parking near the end of the launch: ORT_ENDGAME_JOBS)"""
    scene = gpu_scene("c3_bunny_room")
    w, h, spp, chunk, seed = 333, 187, 24, 2, 5  # 12 chunks
    monkeypatch.setenv("ORT_LPT", "0")
    a, _ = scene.render(w, h, spp, seed, "chunk", chunk=chunk)
    for env in ({}, {"ORT_EXCHANGE": "1"}, {"ORT_EXCHANGE": "1", "ORT_ENDGAME_JOBS": "0"}, {"ORT_EXCHANGE": "1", "ORT_ENDGAME_JOBS": "64"}, {"ORT_EXCHANGE": "0"}):
        monkeypatch.delenv("ORT_LPT", raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        b, _ = scene.render(w, h, spp, seed, "chunk", chunk=chunk)
        assert_bits_equal(a, b, "%s" % (env,))
        r, _ = scene.render(w, h, spp, seed, "chunk", chunk=chunk, rect=(37, 21, 290, 150))
        assert_bits_equal(a[21:150, 37:290], r[21:150, 37:290], "%s, clipped rect" % (env,))
        acc = np.zeros_like(a)
        for k in range(3):
            part, _ = scene.render(w, h, spp, seed, "chunk", chunk=chunk, shard=(k, 3))
            acc += part
        assert_bits_equal(a, acc, "%s, 3-way shard union" % (env,))
        for k in env:
            monkeypatch.delenv(k)


def test_job_batches_do_not_change_a_bit(api, gpu_scene, oracle, monkeypatch):
    """A wave draws its job indices from the job counter in batches and deals them to its lanes (ort_lane.h: draw_job;
    ORT_JOB_BATCH, ORT_BATCH_TAIL).  Which lane runs which job cannot matter: every batch size -- none, odd, larger than
    the job space -- with and without the exact draws near the end, in every kernel family (plain loop at four and five
    waves, ray exchange), PIXEL and CHUNK policies, clipped rects (indices outside the rect are skipped and drawn
    again inside the same call), explicit job lists: equal to the unbatched render and to the oracle, bit for bit."""
    for name, w, h, spp, chunk in (("c3_bunny_room", 333, 187, 24, 2), ("c2_analytic", 203, 117, 16, 4)):
        scene = gpu_scene(name)
        monkeypatch.setenv("ORT_JOB_BATCH", "0")
        a, _ = scene.render(w, h, spp, 9, "chunk", chunk=chunk)
        p0, _ = scene.render(w, h, 4, 9, "pixel")
        monkeypatch.delenv("ORT_JOB_BATCH")
        sc = api.Scene.load_scn(os.path.join(DATA, name + ".scn")).commit()
        ref, _ = oracle.OracleScene(sc.flatten(w, h)).render(w, h, spp, 9, "chunk", chunk=chunk, threads=8)
        assert_bits_equal(a, ref, "%s unbatched vs the oracle" % name)
        for env in ({}, {"ORT_JOB_BATCH": "7"}, {"ORT_JOB_BATCH": "64", "ORT_BATCH_TAIL": "0"}, {"ORT_JOB_BATCH": "1000000"},
                    {"ORT_JOB_BATCH": "33", "ORT_EXCHANGE": "1", "ORT_BATCH_TAIL": "1"}, {"ORT_JOB_BATCH": "128", "ORT_EXCHANGE": "0", "ORT_WAVES5": "1"},
                    {"ORT_JOB_BATCH": "5", "ORT_EXCHANGE": "0", "ORT_WAVES5": "0", "ORT_BATCH_TAIL": "0"}):
            for k, v in env.items():
                monkeypatch.setenv(k, v)
            b, _ = scene.render(w, h, spp, 9, "chunk", chunk=chunk)
            assert_bits_equal(a, b, "%s %s" % (name, env))
            r, _ = scene.render(w, h, spp, 9, "chunk", chunk=chunk, rect=(37, 21, 190, 100))
            assert_bits_equal(a[21:100, 37:190], r[21:100, 37:190], "%s %s, clipped rect" % (name, env))
            p1, _ = scene.render(w, h, 4, 9, "pixel")
            assert_bits_equal(p0, p1, "%s %s, PIXEL policy" % (name, env))
            for k in env:
                monkeypatch.delenv(k)


def test_determinism(api, gpu_scene):
    scene = gpu_scene("testscene")
    a, _ = scene.render(128, 72, 8, 1, "chunk", chunk=4)
    b, _ = scene.render(128, 72, 8, 1, "chunk", chunk=4)
    assert_bits_equal(a, b)


def test_phantom_sphere_hits(api, oracle, gpu_scene):
    """ray_intersect_with_sphere's tangent branch (ray.cpp:174-183) reports t = -b/(2a): a hit in
    mid-air, outside the sphere's box, whose visibility depends on the reference's visiting order.
    The small mirror spheres of the analytic scene produce them; the kernel must agree exactly."""
    scene = gpu_scene("c2_analytic")
    w, h, spp = 160, 160, 16
    img, st = scene.render(w, h, spp, 99, "chunk", chunk=4, counters=True)
    ref, _ = oracle.OracleScene(scene.flatten(w, h)).render(w, h, spp, 99, "chunk", chunk=4, threads=16)
    assert st["fallback_rays"] > 0
    assert_bits_equal(img, ref)


def test_exactness_fallback_is_exercised(api, oracle, gpu_scene):
    """some bounce origins land exactly on a reference octree node face; the kernel must then
    reproduce the reference's culling (Exactness).  Large enough to hit the case."""
    scene = gpu_scene("c4_dwarf_room")
    w, h, spp = 160, 120, 16
    img, st = scene.render(w, h, spp, 7, "chunk", chunk=4, counters=True)
    ref, _ = oracle.OracleScene(scene.flatten(w, h)).render(w, h, spp, 7, "chunk", chunk=4, threads=16)
    assert st["fallback_rays"] > 0
    assert_bits_equal(img, ref)


@pytest.mark.parametrize("name", ["testscene", "c3_bunny_room", "glass_room", "c5_heightfield_224"])
def test_every_ray_through_the_reference_order_walk(api, oracle, gpu_scene, monkeypatch, name):
    """ORT_DEBUG_FORCE_FALLBACK=0 sends EVERY ray through the breadth-first emulation of raycast_bvh
    (ray.cpp:624-822) on the device: same image, and thousands of lanes contend for the pooled queues;
    mask 0xf re-casts ~1/16 of the rays, mixing both walks inside a wave."""
    scene = gpu_scene(name)
    w, h, spp = 64, 40, 4
    ref, ost = oracle.OracleScene(scene.flatten(w, h)).render(w, h, spp, 77, "chunk", chunk=2, threads=16)
    monkeypatch.setenv("ORT_DEBUG_FORCE_FALLBACK", "0")
    img, st = scene.render(w, h, spp, 77, "chunk", chunk=2, counters=True)
    assert st["fallback_rays"] == st["rays"] == ost["rays"]
    assert_bits_equal(img, ref, "all rays re-cast")
    monkeypatch.setenv("ORT_DEBUG_FORCE_FALLBACK", "0xf")
    img, st = scene.render(w, h, spp, 77, "chunk", chunk=2, counters=True)
    assert 0 < st["fallback_rays"] < st["rays"]
    assert_bits_equal(img, ref, "1/16 of the rays re-cast")


@pytest.mark.parametrize("name,n_jobs", [("c3_bunny_room", 1), ("testscene", 3), ("c5_heightfield_224", 7)])
def test_reference_order_walk_with_few_lanes(api, oracle, gpu_scene, monkeypatch, name, n_jobs):
    """The exact fallback is walked by the lanes that stand in resolve_hit together (ref_raycast_bfs_wave): records
    and children of a node are dealt to them by rank.  One, three and seven jobs = that many lanes with a path: the
    dealing wraps around (eight children over fewer lanes).  Every ray forced through the walk; image and final
    RandomSeries states against the oracle."""
    scene = gpu_scene(name)
    monkeypatch.setenv("ORT_DEBUG_FORCE_FALLBACK", "0")
    w, h = 48, 32
    jobs = np.zeros(n_jobs, api.JOB_DTYPE)
    for i in range(n_jobs):
        x0, y0 = 6 * i, 4 * i
        jobs[i] = (x0, y0, x0 + 5, y0 + 3, 1000 + 17 * i, 2 + (i % 3))
    out = np.zeros((h, w, 3), "<f4")
    finals, _ = scene.tiled_raytrace_batch(out, jobs)
    ref = np.zeros((h, w, 3), "<f4")
    osc = oracle.OracleScene(scene.flatten(w, h))
    for i, j in enumerate(jobs):
        _, st = osc.tiled_raytrace(ref, int(j["x0"]), int(j["y0"]), int(j["x1"]), int(j["y1"]), int(j["rng_state"]), int(j["spp"]))
        assert finals[i] == st
    assert_bits_equal(out, ref, name)


@pytest.mark.parametrize("name,w,h,spp,seed,policy,chunk,x,y", [
    ("c2_analytic", 480, 270, 32, 7066, "pixel", 0, 25, 100),
    ("letters", 480, 270, 32, 7092, "chunk", 8, 8, 265),
    ("c2_analytic", 480, 270, 32, 7186, "pixel", 0, None, None),
    ("testscene", 256, 144, 16, 7425, "chunk", 8, None, None)])
def test_hits_in_front_of_their_node_box(api, oracle, gpu_scene, name, w, h, spp, seed, policy, chunk, x, y):
    """found by tools/stress_parity.py (7 single-pixel differences in 2.5e9 paths): the reference's cylinder
    boxes are too small (r * (1 - |a_k|/|a|) instead of r * sqrt(1 - a_k^2)) and flat boxes round either way,
    so a hit can lie in FRONT of its node's entry distance; the reference then sees it only if nothing
    nearer than that entry was found earlier.  The chain check bounds the entry by the winner's distance or,
    failing that, by the runner-up; else the ray is re-cast exactly."""
    scene = gpu_scene(name)
    rect = (0, 0, w, h) if x is None else (max(x - 3, 0), max(y - 3, 0), min(x + 4, w), min(y + 4, h))
    img, _ = scene.render(w, h, spp, seed, policy, chunk=chunk, rect=rect)
    ref, _ = oracle.OracleScene(scene.flatten(w, h)).render(w, h, spp, seed, policy, chunk=max(chunk, 1), rect=rect, threads=16)
    assert_bits_equal(img[rect[1]:rect[3], rect[0]:rect[2]], ref[rect[1]:rect[3], rect[0]:rect[2]], name)


def test_quadric_hits_reported_outside_the_shape(api, oracle, gpu_scene):
    """found by tools/stress_parity.py (testscene, rr 0.95, seed 150229): 21 units from an r = 0.05 cylinder the
    reference's f32 quadratic reports a hit 2.8e-4 OUTSIDE the cylinder, on a ray that misses a tight box around
    it.  The fast tree's sphere and cylinder boxes therefore carry the intersector's error bound
    (ort_tree.cpp); this is the pixel that differed."""
    scene = gpu_scene("testscene")
    w, h, spp, seed, rr = 256, 144, 16, 150229, 0.95
    rect = (200, 100, 218, 115)
    img, _ = scene.render(w, h, spp, seed, "chunk", chunk=8, rect=rect, rr=rr)
    ref, _ = oracle.OracleScene(scene.flatten(w, h)).render(w, h, spp, seed, "chunk", chunk=8, rect=rect, rr=rr, threads=16)
    assert_bits_equal(img[rect[1]:rect[3], rect[0]:rect[2]], ref[rect[1]:rect[3], rect[0]:rect[2]])


def test_slanted_cylinders_are_settled_without_the_breadth_first_walk(api, oracle, gpu_scene):
    """rand_b holds cylinders with oblique axes, whose reference node boxes (r * (1 - |a_k|/|a|) per axis) leave
    much of the cylinder outside: 0.8 % of all rays hit a cylinder in front of, or entirely outside, its box.
    Those rays are settled by one more fast traversal (the winner ignored; resolve_hit), not by the
    single-lane breadth-first emulation -- which made this scene 80x slower before."""
    scene = gpu_scene("rand_b")
    w, h, spp = 192, 108, 16
    img, st = scene.render(w, h, spp, 7, "chunk", chunk=4, counters=True)
    ref, _ = oracle.OracleScene(scene.flatten(w, h)).render(w, h, spp, 7, "chunk", chunk=4, threads=16)
    assert_bits_equal(img, ref)
    assert st["fallback_rays"] * 5000 < st["rays"], (st["fallback_rays"], st["rays"])


# ---- synthetic scenes / edge cases -----------------------------------------------------------------
def _random_scene(api, seed, n_sph=6, n_box=4, n_cyl=3, n_tri=40):
    rng = np.random.default_rng(seed)
    mats = np.zeros(7, api.MATERIAL_DTYPE)
    mats["diffuse"][1] = (0.7, 0.7, 0.7)
    mats["diffuse"][2] = (0.2, 0.6, 0.3); mats["specular"][2, :3] = 1
    mats["specular"][3, :3] = 1
    mats["transmission"][4] = 1; mats["ior"][4] = 1.4
    mats["is_light"][5] = 1; mats["emit"][5] = (3, 3, 2)
    mats["diffuse"][6] = (0.5, 0.1, 0.1); mats["transmission"][6] = (0.3, 0.3, 0.3); mats["ior"][6] = 1.2
    mats["ior"][1:4] = 1.0
    # closed room made of six slabs (avoids the reference's undefined primary miss)
    boxes = np.zeros(6 + n_box, api.BOX_DTYPE)
    room = [((-4, -4, -0.2), (4, 4, 0)), ((-4, -4, 5), (4, 4, 5.2)), ((-4.2, -4, -0.2), (-4, 4, 5.2)),
            ((4, -4, -0.2), (4.2, 4, 5.2)), ((-4, -4.2, -0.2), (4, -4, 5.2)), ((-4, 4, -0.2), (4, 4.2, 5.2))]
    for i, (lo, hi) in enumerate(room):
        boxes[i] = (lo, hi, 1 + (i % 2))
    for i in range(n_box):
        lo = rng.uniform(-3, 2.5, 3); lo[2] = rng.uniform(0, 2)
        boxes[6 + i] = (lo, lo + rng.uniform(0.2, 1.2, 3), rng.integers(1, 5))
    sph = np.zeros(n_sph + 1, api.SPHERE_DTYPE)
    for i in range(n_sph):
        sph[i] = (rng.uniform(-3, 3, 3) * (1, 1, 0) + (0, 0, rng.uniform(0.3, 3)), rng.uniform(0.2, 0.7), rng.integers(1, 7))
    sph[n_sph] = ((0, 0, 4.2), 0.7, 5)  # the light
    cyl = np.zeros(n_cyl, api.CYLINDER_DTYPE)
    for i in range(n_cyl):
        axis = rng.normal(size=3) * rng.uniform(0.5, 2)
        if i == 0:
            axis = (0, 0, 1.5)
        cyl[i] = (rng.uniform(-2.5, 2.5, 3) * (1, 1, 0) + (0, 0, 0.5), axis, rng.uniform(0.1, 0.3), rng.integers(1, 5))
    lights = np.array([(2, i) for i in range(n_cyl)] + [(1, n_sph)], api.LIGHT_DTYPE)
    verts = (rng.uniform(-2.5, 2.5, size=(n_tri * 3, 3)) * (1, 1, 0.4) + (0, 0, 1.2)).astype("<f4")
    verts[1::3] = verts[0::3] + rng.normal(scale=0.5, size=(n_tri, 3))
    verts[2::3] = verts[0::3] + rng.normal(scale=0.5, size=(n_tri, 3))
    idx = np.arange(n_tri * 3, dtype="<u4")
    mesh = dict(vertices=verts, indices=idx, mat=int(rng.integers(1, 5)))
    q = np.array([0.416981 * 0 + 0.279589, 0.480987, 0.718247, 0.416981], "<f4")  # xyzw of testscene's camera
    return api.Scene.from_arrays(mats, sph, boxes, cyl, lights, [mesh] if n_tri else [], camera_p=(3.3, 2.0, 2.6),
                                 camera_quat_xyzw=q, camera_height_ratio=0.3, screen=(64, 48))


@pytest.mark.parametrize("seed", range(6))
def test_random_scenes(api, oracle, seed):
    """random spheres/boxes/cylinders/triangles with diffuse, mirror, glass, coloured-glass and
    emissive materials; the oracle is fed the same arrays through the ABI getters."""
    scene = _random_scene(api, seed).commit().upload(0)
    w, h, spp = 48, 36, 8
    img, st = scene.render(w, h, spp, 1000 + seed, "chunk", chunk=4, counters=True)
    ref, _ = oracle.OracleScene(scene.flatten(w, h), with_reference_csg=False).render(w, h, spp, 1000 + seed, "chunk", chunk=4, threads=16)
    assert_bits_equal(img, ref, "random scene %d" % seed)
    scene.close()


def test_no_mesh_and_single_primitive(api, oracle):
    scene = _random_scene(api, 99, n_tri=0).commit().upload(0)
    img, _ = scene.render(33, 21, 3, 5, "pixel")
    ref, _ = oracle.OracleScene(scene.flatten(33, 21), with_reference_csg=False).render(33, 21, 3, 5, "pixel", threads=8)
    assert_bits_equal(img, ref)
    scene.close()
    # one emissive sphere around the camera: every primary ray hits the light from inside
    mats = np.zeros(2, api.MATERIAL_DTYPE)
    mats["is_light"][1] = 1
    mats["emit"][1] = (1, 2, 3)
    sph = np.zeros(1, api.SPHERE_DTYPE)
    sph[0] = ((0, 0, 0), 50.0, 1)
    lights = np.array([(1, 0)], api.LIGHT_DTYPE)
    s = api.Scene.from_arrays(mats, sph, lights=lights, camera_p=(1, 1, 1), camera_height_ratio=0.2).commit().upload(0)
    img, _ = s.render(16, 8, 4, 3, "pixel")
    assert (img == np.array([1, 2, 3], "<f4")).all()
    s.close()


def _degenerate_scene(api, variant):
    """closed room plus geometry chosen to hit ties and degenerate branches"""
    mats = np.zeros(7, api.MATERIAL_DTYPE)
    mats["diffuse"][1] = (0.7, 0.7, 0.7); mats["diffuse"][2] = (0.8, 0.2, 0.2); mats["diffuse"][3] = (0.2, 0.2, 0.8)
    mats["specular"][4, :3] = 1
    mats["transmission"][5] = 1; mats["ior"][5] = 1.5
    mats["is_light"][6] = 1; mats["emit"][6] = (4, 4, 4)
    mats["ior"][1:5] = 1.0
    room = [((-4, -4, -0.2), (4, 4, 0)), ((-4, -4, 5), (4, 4, 5.2)), ((-4.2, -4, -0.2), (-4, 4, 5.2)),
            ((4, -4, -0.2), (4.2, 4, 5.2)), ((-4, -4.2, -0.2), (4, -4, 5.2)), ((-4, 4, -0.2), (4, 4.2, 5.2))]
    boxes = [(lo, hi, 1) for lo, hi in room]
    spheres = [((0, 0, 4.2), 0.7, 6)]
    cyls = []
    verts, idx = [], []
    if variant == 0:
        # coincident boxes with different materials (bit-equal t: first tested wins), a box face in the floor plane
        boxes += [((-1, -1, 0), (1, 1, 1), 2), ((-1, -1, 0), (1, 1, 1), 3), ((1, -1, 0), (2, 1, 1), 3), ((-0.5, -0.5, 1), (0.5, 0.5, 1.5), 4)]
        # coincident spheres, concentric spheres (glass shell around a diffuse core), sphere touching a box face
        spheres += [((2, 2, 1), 0.5, 2), ((2, 2, 1), 0.5, 3), ((-2, 2, 1), 0.8, 5), ((-2, 2, 1), 0.4, 2), ((0, -2.5, 0.5), 0.5, 4)]
    elif variant == 1:
        # zero-area and duplicated triangles, a quad as two fan triangles plus its mirror image (same plane, opposite winding)
        quad = [(-1, -1, 1), (1, -1, 1), (1, 1, 1), (-1, 1, 1)]
        verts += quad + quad[::-1] + [(0, 0, 2), (0, 0, 2), (0, 0, 2)] + [(2, 0, 0.5), (3, 0, 0.5), (4, 0, 0.5)]
        idx += [0, 1, 2, 0, 2, 3, 4, 5, 6, 4, 6, 7, 8, 9, 10, 11, 12, 13, 0, 1, 2]
        cyls += [((-2, -2, 0), (0, 0, 2), 0.3, 2), ((-2, -2, 0), (0, 0, 2), 0.3, 3), ((2, -2, 1), (0, 0, -1), 0.2, 4),
                 ((0, 3, 1), (2, 0, 0), 1e-3, 2)]
    else:
        # shapes poking through the walls, a sphere bigger than the room's half, a cylinder along a wall plane
        spheres += [((4, 0, 2), 1.0, 2), ((0, 0, 0), 1.0, 5), ((-3.5, -3.5, 0.5), 0.5, 4)]
        boxes += [((3.5, 3.5, 0), (4.5, 4.5, 6), 3)]
        cyls += [((-4, -3, 1), (0, 6, 0), 0.25, 2), ((0, 0, 0), (0, 0, 5), 0.1, 4)]
    sph = np.zeros(len(spheres), api.SPHERE_DTYPE)
    for i, v in enumerate(spheres):
        sph[i] = v
    box = np.zeros(len(boxes), api.BOX_DTYPE)
    for i, v in enumerate(boxes):
        box[i] = v
    cyl = np.zeros(len(cyls), api.CYLINDER_DTYPE)
    for i, v in enumerate(cyls):
        cyl[i] = v
    lights = np.array([(2, i) for i in range(len(cyls))] + [(1, 0)], api.LIGHT_DTYPE)
    meshes = [dict(vertices=np.array(verts, "<f4"), indices=np.array(idx, "<u4"), mat=2)] if verts else []
    q = np.array([0.279589, 0.480987, 0.718247, 0.416981], "<f4")
    return api.Scene.from_arrays(mats, sph, box, cyl, lights, meshes, camera_p=(3.3, 2.0, 2.6), camera_quat_xyzw=q,
                                 camera_height_ratio=0.3, screen=(64, 48))


@pytest.mark.parametrize("variant", range(3))
def test_degenerate_geometry(api, oracle, variant):
    """coincident / concentric / touching shapes (ties in t), zero-area and duplicated triangles, a quad and its
    mirror image, needle cylinders, shapes that poke through the room: same bits as the oracle"""
    scene = _degenerate_scene(api, variant).commit().upload(0)
    w, h, spp = 64, 48, 16
    img, st = scene.render(w, h, spp, 4000 + variant, "chunk", chunk=4, counters=True)
    ref, ost = oracle.OracleScene(scene.flatten(w, h), with_reference_csg=False).render(w, h, spp, 4000 + variant, "chunk", chunk=4, threads=16)
    assert st["rays"] == ost["rays"]
    assert_bits_equal(img, ref, "degenerate scene %d" % variant)
    scene.close()


def test_open_scene_terminates(api):
    """primary misses are undefined behaviour in the reference; defined here as
    'the path ends with zero radiance'.  Must not hang or fault."""
    mats = np.zeros(3, api.MATERIAL_DTYPE)
    mats["diffuse"][1] = 0.5
    mats["is_light"][2] = 1
    mats["emit"][2] = (2, 2, 2)
    box = np.zeros(1, api.BOX_DTYPE)
    box[0] = ((-5, -5, -0.1), (5, 5, 0), 1)
    sph = np.zeros(1, api.SPHERE_DTYPE)
    sph[0] = ((0, 0, 3), 0.5, 2)
    s = api.Scene.from_arrays(mats, sph, box, lights=np.array([(1, 0)], api.LIGHT_DTYPE), camera_p=(4, 3, 2.5),
                              camera_quat_xyzw=(0.279589, 0.480987, 0.718247, 0.416981), camera_height_ratio=0.3)
    s.commit().upload(0)
    img, st = s.render(64, 48, 8, 1, "chunk", chunk=4, counters=True)
    assert np.isfinite(img).all()
    assert st["paths"] == 64 * 48 * 8
    s.close()


def test_cli_scn_to_hdr_end_to_end(api, oracle, gpu_scene, tmp_path):
    """bin/ort_render: .scn in, .hdr out -- the file equals the oracle's RGBE encoding of the oracle's image,
    byte for byte, and the raw dump equals the image"""
    import subprocess
    cli = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "offline_raytracer_amd", "bin", "ort_render")
    w, h, spp = 64, 36, 8
    out, raw = str(tmp_path / "o.hdr"), str(tmp_path / "o.f32")
    for policy, chunk in (("chunk", 4), ("tile32", 0)):
        args = [cli, "--scene", os.path.join(os.path.dirname(GOLDEN), "..", "data", "c4_dwarf_room.scn"), "--width", str(w), "--height", str(h),
                "--spp", str(spp), "--seed", "31", "--policy", policy, "--out", out, "--raw", raw]
        if chunk:
            args += ["--chunk", str(chunk)]
        r = subprocess.run(args, capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        ref, _ = oracle.OracleScene(gpu_scene("c4_dwarf_room").flatten(w, h)).render(w, h, spp, 31, policy, chunk=max(chunk, 1), threads=16)
        assert_bits_equal(np.fromfile(raw, "<f4").reshape(h, w, 3), ref, "raw dump, %s" % policy)
        want = str(tmp_path / "want.hdr")
        assert oracle.lib().oracle_write_hdr(want.encode(), np.ascontiguousarray(ref).ctypes.data, w, h) == 0
        assert open(out, "rb").read() == open(want, "rb").read(), policy


def test_determinant_threshold_scene(api, oracle, manifest, tmp_path_factory):
    """DESIGN.md section 3's residual, constructed (tools/make_detscene.py): tiny triangles whose |e1 x e2| straddles the
    1e-6 determinant threshold of ray.cpp:95-96 -- half the patch exists for the reference, half does not, the border is
    decided by the last bits of det -- and a strip seen edge-on through the aperture ring, where hits exist only by
    rounding.  The kernel must return the reference's own pixels (golden, from the compiled reference) and the
    oracle's on more samples and other seeds."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import make_detscene
    d = str(tmp_path_factory.mktemp("detscene")) + "/"
    scn, nv, nf = make_detscene.write_scene(d)
    meta = manifest["detscene"]
    assert (nv, nf) == (meta["vertices"], meta["triangles"])
    scene = api.Scene.load_scn(scn).commit().upload(0)
    W, H = meta["width"], meta["height"]
    z = np.load(os.path.join(GOLDEN, "renders_detscene.npz"))
    for c in meta["cases"]:
        x0, y0, x1, y1 = meta["windows"][c["window"]]
        img, _ = scene.render(W, H, c["spp"], c["seed"], c["policy"], chunk=c["chunk"], rect=(x0, y0, x1, y1))
        assert_bits_equal(img[y0:y1, x0:x1], z[c["key"]], "reference golden " + c["key"])
    osc = oracle.OracleScene(scene.flatten(W, H))
    for name, seed, spp, chunk in (("patchA", 1, 32, 8), ("stripB", 2, 32, 8), ("patchA", 3, 16, 16)):
        x0, y0, x1, y1 = meta["windows"][name]
        img, st = scene.render(W, H, spp, seed, "chunk", chunk=chunk, rect=(x0, y0, x1, y1), counters=True)
        ref, _ = osc.render(W, H, spp, seed, "chunk", chunk=chunk, rect=(x0, y0, x1, y1), threads=16)
        assert_bits_equal(img[y0:y1, x0:x1], ref[y0:y1, x0:x1], "oracle %s seed %d" % (name, seed))
        assert st["tri_tests"] > 0
    scene.close()
