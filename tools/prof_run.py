"""Developer script for rocprofv3 runs: renders the bench workload at a reduced spp a few times.
usage: python3 tools/prof_run.py [scene] [W] [H] [spp] [chunk] [reps]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from offline_raytracer_amd import api
a = sys.argv[1:]
name = a[0] if len(a) > 0 else "c3_bunny_room"
W = int(a[1]) if len(a) > 1 else 1920
H = int(a[2]) if len(a) > 2 else 1080
spp = int(a[3]) if len(a) > 3 else 64
chunk = int(a[4]) if len(a) > 4 else 64
reps = int(a[5]) if len(a) > 5 else 2
if name.startswith("c5:"):  # generated height field, e.g. c5:708
    import tempfile
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import make_heightfield
    path, _, _ = make_heightfield.write_scene(int(name[3:]), tempfile.mkdtemp(prefix="c5_"))
else:
    path = os.path.join(ROOT, "data", name + ".scn")
scene = api.Scene.load_scn(path).commit().upload(0)
for i in range(reps):
    img, st = scene.render(W, H, spp, 12345, "chunk", chunk=chunk)
    print("rep", i, "kernel_ms %.2f -> %.1f Mpaths/s" % (st["kernel_ms"], W * H * spp / st["kernel_ms"] / 1e3), flush=True)
