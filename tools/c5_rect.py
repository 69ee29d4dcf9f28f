import os, sys, tempfile, time
ROOT = "/root/repo" if os.path.exists("/root/repo/tools") else os.environ["GRAFT_REPO_ROOT"]
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import make_heightfield
from offline_raytracer_amd import api
d = tempfile.mkdtemp(prefix="c5_")
scn, nv, nf = make_heightfield.write_scene(708, d)
scene = api.Scene.load_scn(scn).commit().upload(0)
W, H = 3840, 2160
spp = int(sys.argv[1])
for frac in (8, 4, 2, 1):
    y1 = H // frac
    for rep in range(2):
        img, st = scene.render(W, H, spp, 12345, "chunk", chunk=spp, rect=(0, 0, W, y1))
    print("rows 0..%d: kernel_ms %.1f -> %.1f Mpaths/s" % (y1, st["kernel_ms"], W * y1 * spp / st["kernel_ms"] / 1e3), flush=True)
# and the top part of the image alone
img, st = scene.render(W, H, spp, 12345, "chunk", chunk=spp, rect=(0, H // 2, W, H))
print("rows %d..%d: kernel_ms %.1f -> %.1f Mpaths/s" % (H // 2, H, st["kernel_ms"], W * (H - H // 2) * spp / st["kernel_ms"] / 1e3), flush=True)
