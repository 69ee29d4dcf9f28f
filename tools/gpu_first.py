"""Developer script: first contact with the GPU (parity + rough timing)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from offline_raytracer_amd import api
import oracle_lib

print("devices", api.device_count())
for name in ["c2_analytic", "c3_bunny_room", "c4_dwarf_room", "testscene"]:
    scene = api.Scene.load_scn(os.path.join(ROOT, "data", name + ".scn")).commit().upload(0)
    print(name, scene.tree_info())
    W = H = 96
    for policy, spp, chunk in [("chunk", 8, 4), ("pixel", 4, 0), ("tile32", 2, 0), ("whole", 1, 0)]:
        img, st = scene.render(W, H, spp, 7, policy, chunk=chunk, counters=True)
        ref, ost = oracle_lib.OracleScene(scene.flatten(W, H)).render(W, H, spp, 7, policy, chunk=max(chunk, 1), threads=16)
        same = np.array_equal(img.view("<u4"), ref.view("<u4"))
        print("  ", policy, "bit-identical" if same else "DIFF %d px" % int((img.view("<u4") != ref.view("<u4")).any(2).sum()), st)
    for (W, H, spp, chunk) in [(480, 270, 64, 16), (1920, 1080, 16, 16), (1920, 1080, 64, 64)]:
        t0 = time.time()
        img, st = scene.render(W, H, spp, 7, "chunk", chunk=chunk)
        dt = time.time() - t0
        print("   %dx%d %dspp: kernel %.1f ms -> %.1f Mpaths/s (wall %.2fs) mean %s" % (W, H, spp, st["kernel_ms"], W * H * spp / st["kernel_ms"] / 1e3, dt, img.mean(axis=(0, 1))))
    scene.close()
