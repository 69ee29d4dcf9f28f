"""Developer script: one-off parity hunt on the GPU box -- many random scenes and larger renders of the
fixture scenes, HIP path vs oracle, bit for bit.  Prints one line per case; exits 1 on any difference.
usage: python3 tools/stress_parity.py [n_random=40] [budget_seconds=240]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from offline_raytracer_amd import api
import oracle_lib
import test_gpu_parity as T

COUNTERS = os.environ.get("STRESS_COUNTERS", "1") != "0"  # 0: the production kernel flavours (no work counters; ORT_EXCHANGE=1 forces the exchange loop)
n_random = int(sys.argv[1]) if len(sys.argv) > 1 else 40
budget = float(sys.argv[2]) if len(sys.argv) > 2 else 240.0
t_start = time.time()
bad = 0


def check(tag, scene, w, h, spp, seed, policy, chunk, csg):
    global bad
    rr = (0.8, 0.8, 0.8, 0.5, 0.95)[seed % 5]  # the reference's literal mostly; the parameter's range sometimes
    tag = tag if rr == 0.8 else "%s rr%.2f" % (tag, rr)
    img, st = scene.render(w, h, spp, seed, policy, chunk=chunk, counters=COUNTERS, rr=rr)
    ref, ost = oracle_lib.OracleScene(scene.flatten(w, h), with_reference_csg=csg).render(w, h, spp, seed, policy, chunk=max(chunk, 1), rr=rr, threads=16)
    diff = int((img.view("<u4") != ref.view("<u4")).any(axis=2).sum())
    bad += diff != 0
    print("%-34s %4dx%-4d %3dspp %-5s c%-2d seed %-6d paths %9d fallback %6d : %s" %
          (tag, w, h, spp, policy, chunk, seed, st["paths"], st["fallback_rays"], "ok" if diff == 0 else "DIFF in %d pixels" % diff), flush=True)


rng = np.random.default_rng(2024)
for i in range(n_random):
    if time.time() - t_start > budget * 0.5:
        break
    seed = 500 + i
    sc = T._random_scene(api, seed, n_sph=int(rng.integers(0, 12)), n_box=int(rng.integers(0, 8)), n_cyl=int(rng.integers(1, 6)),
                         n_tri=int(rng.integers(0, 400))).commit().upload(0)
    check("random scene %d" % seed, sc, 96, 72, 16, seed, "chunk", 4, False)
    sc.close()
cases = [("c2_analytic", 480, 270, 32), ("c4_dwarf_room", 480, 270, 32), ("letters", 480, 270, 32), ("glass_room", 480, 270, 32),
         ("c3_bunny_room", 320, 180, 32), ("testscene", 256, 144, 16), ("c5", 240, 135, 8)]
seed0 = int(os.environ.get("STRESS_SEED0", "7000"))
scenes = {}
for name, _, _, _ in cases:
    if name == "c5":
        import tempfile
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import make_heightfield
        path, _, _ = make_heightfield.write_scene(224, tempfile.mkdtemp(prefix="c5_"))
    else:
        path = os.path.join(ROOT, "data", name + ".scn")
    scenes[name] = api.Scene.load_scn(path).commit().upload(0)
def check_batch(name, sc, seed):
    """the batch form of the single call: random disjoint rects (some empty, some one pixel), random nonzero
    RNG states and sample counts, one lane per job; image and every final RNG state against the oracle"""
    global bad
    r = np.random.default_rng(seed)
    w, h = 96, 64
    cells = [(cx, cy) for cy in range(0, h, 16) for cx in range(0, w, 16)]
    r.shuffle(cells)
    jobs = np.zeros(len(cells), api.JOB_DTYPE)
    for i, (cx, cy) in enumerate(cells):
        x0 = cx + int(r.integers(0, 8)); y0 = cy + int(r.integers(0, 8))
        x1 = x0 + int(r.integers(0, 9)); y1 = y0 + int(r.integers(0, 9))  # width / height 0 => empty rect
        jobs[i] = (x0, y0, min(x1, w), min(y1, h), int(r.integers(1, 2 ** 32)), int(r.integers(1, 5)))
    out = np.zeros((h, w, 3), "<f4")
    finals, _ = sc.tiled_raytrace_batch(out, jobs)
    ref = np.zeros((h, w, 3), "<f4")
    osc = oracle_lib.OracleScene(sc.flatten(w, h))
    states_ok = True
    for i, j in enumerate(jobs):
        _, st = osc.tiled_raytrace(ref, int(j["x0"]), int(j["y0"]), int(j["x1"]), int(j["y1"]), int(j["rng_state"]), int(j["spp"]))
        states_ok &= (finals[i] == st)
    diff = int((out.view("<u4") != ref.view("<u4")).any(axis=2).sum())
    bad += (diff != 0) or (not states_ok)
    print("%-34s batch of %d jobs seed %-6d : %s" % (name, len(jobs), seed, "ok" if diff == 0 and states_ok else "DIFF px %d states %s" % (diff, states_ok)), flush=True)


k = 0
while time.time() - t_start < budget:
    name, w, h, spp = cases[k % len(cases)]
    sc = scenes[name]
    if k % 5 == 4 and name != "c5":
        check_batch(name, sc, seed0 + k)
    if k % 11 == 10:  # the reference's own schedules: long serial jobs
        check(name, sc, 96, 64, 2, seed0 + k, "tile32" if k % 2 else "whole", 0, True)
    else:
        check(name, sc, w, h, spp, seed0 + k, "chunk" if k % 3 else "pixel", 8 if k % 3 else 0, True)
    k += 1
print("cases with differences:", bad)
sys.exit(1 if bad else 0)
