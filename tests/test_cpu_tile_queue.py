"""SURVEY 8 row f4 on Linux: the reference's driver -- main()'s 1 024 tiles handed to a pool of worker threads through a work
queue (macos_main.mm:565-598, 602-662) -- as a developer tool, tools/host_sim: the kernel's own lane code (ort_lane.h) compiled
for the host, one simulated lane per thread, all workers on one job counter.  It is NOT a fallback of the product (the library
never loads or links it; the render call has no CPU path): here it is held against the pixels of the compiled reference
(golden fixtures) and against the oracle, bit for bit, single- and multi-threaded."""
import os
import subprocess

import numpy as np
import pytest

from conftest import DATA, GOLDEN, ROOT, assert_bits_equal

TOOL = os.path.join(ROOT, "tools", "host_sim")


@pytest.fixture(scope="module")
def host_sim():
    src = [os.path.join(ROOT, "tools", "host_sim.cpp"), os.path.join(ROOT, "offline_raytracer_amd", "csrc", "ort_lane.h")]
    if not os.path.exists(TOOL) or any(os.path.getmtime(s) > os.path.getmtime(TOOL) for s in src):
        if not os.path.exists(os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")):
            pytest.skip("tools/host_sim is not built and there is no hipcc to build it with")
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tools")], stderr=subprocess.DEVNULL)
    return TOOL


def _run(tool, scene, w, h, spp, seed, policy, chunk, out, threads, extra=None):
    env = dict(os.environ, SIM_THREADS=str(threads))
    env.update(extra or {})
    r = subprocess.run([tool, os.path.join(DATA, scene + ".scn"), DATA + "/", str(w), str(h), str(spp), str(seed), policy, str(chunk), out],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-1500:]
    return np.fromfile(out, "<f4").reshape(h, w, 3)


def test_tile_queue_on_threads_reproduces_the_reference_schedule(host_sim, tmp_path):
    """main()'s own schedule (TILE32: 32x32 tiles, tile seeds drawn from the master series in row-major order) on 8 worker
    threads == on one == the compiled reference's pixels (tests/golden/renders_testscene.npz, made by oracle/_ref/ref_det)"""
    z = np.load(os.path.join(GOLDEN, "renders_testscene.npz"))
    want = z["tile32_64x64_2spp_c1_s12345"]
    one = _run(host_sim, "testscene", 64, 64, 2, 12345, "tile32", 0, str(tmp_path / "t1.f32"), 1)
    many = _run(host_sim, "testscene", 64, 64, 2, 12345, "tile32", 0, str(tmp_path / "t8.f32"), 8)
    assert_bits_equal(one, want, "one worker vs the reference")
    assert_bits_equal(many, want, "eight workers vs the reference")


@pytest.mark.parametrize("scene,w,h,spp,policy,chunk,extra", [
    ("c2_analytic", 93, 61, 6, "chunk", 2, {}),            # ragged edge blocks, all lobes
    ("c3_bunny_room", 96, 64, 4, "pixel", 0, {"SIM_DIFFUSE": "1"}),
    ("c3_bunny_room", 96, 64, 4, "chunk", 2, {"SIM_WIDE": "1"}),  # the 4-wide tree
])
def test_worker_pool_matches_the_oracle(host_sim, api, oracle, tmp_path, scene, w, h, spp, policy, chunk, extra):
    got = _run(host_sim, scene, w, h, spp, 77, policy, chunk, str(tmp_path / "o.f32"), 8, extra)
    sc = api.Scene.load_scn(os.path.join(DATA, scene + ".scn")).commit()
    ref, _ = oracle.OracleScene(sc.flatten(w, h)).render(w, h, spp, 77, policy, chunk=max(chunk, 1), threads=8)
    assert_bits_equal(got, ref, "%s %s on 8 workers vs the oracle" % (scene, policy))
