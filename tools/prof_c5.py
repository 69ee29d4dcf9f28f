"""Developer script: time the generated C5 scene (1M-triangle height field) at 3840x2160.
usage: python3 tools/prof_c5.py [n=708] [spp=64] [chunk=64] [reps=2]"""
import os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import make_heightfield
from offline_raytracer_amd import api
a = sys.argv[1:]
n = int(a[0]) if len(a) > 0 else 708
spp = int(a[1]) if len(a) > 1 else 64
chunk = int(a[2]) if len(a) > 2 else 64
reps = int(a[3]) if len(a) > 3 else 2
W, H = 3840, 2160
d = tempfile.mkdtemp(prefix="c5_")
t0 = time.time(); scn, nv, nf = make_heightfield.write_scene(n, d); t1 = time.time()
scene = api.Scene.load_scn(scn); t2 = time.time()
scene.commit(); t3 = time.time()
scene.upload(0); t4 = time.time()
print("generate %.2fs parse %.2fs commit %.2fs upload %.2fs; %d triangles; tree %s" % (t1 - t0, t2 - t1, t3 - t2, t4 - t3, nf, scene.tree_info()), flush=True)
for i in range(reps):
    img, st = scene.render(W, H, spp, 12345, "chunk", chunk=chunk)
    print("rep", i, "fallback_rays", st.get("fallback_rays"), "kernel_ms %.2f -> %.1f Mpaths/s" % (st["kernel_ms"], W * H * spp / st["kernel_ms"] / 1e3), flush=True)
img, st = scene.render(W, H, min(spp, 8), 12345, "chunk", chunk=min(chunk, 8), counters=True)
R = st["rays"] / st["paths"]; Vn = st["node_tests"] / st["rays"]; Vt = st["tri_tests"] / st["rays"]; Vp = st["analytic_tests"] / st["rays"]
print("rays/path %.2f node/ray %.1f tri/ray %.2f analytic/ray %.2f -> %.0f B/path; fallback rays %d" % (R, Vn, Vt, Vp, R * (Vn * 32 + Vt * 36 + Vp * 32), st["fallback_rays"]))
