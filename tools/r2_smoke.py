"""Developer script (GPU box): small renders against the oracle, exchange on and off."""
import os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from offline_raytracer_amd import api
import oracle_lib
for name, W, H, spp, chunk in (("c3_bunny_room", 64, 48, 8, 4), ("testscene", 96, 64, 8, 4), ("c3_bunny_room", 320, 200, 16, 8)):
    scene = api.Scene.load_scn(os.path.join(ROOT, "data", name + ".scn")).commit().upload(0)
    img, st = scene.render(W, H, spp, 2024, "chunk", chunk=chunk)
    ref, _ = oracle_lib.OracleScene(scene.flatten(W, H)).render(W, H, spp, 2024, "chunk", chunk=chunk, threads=8)
    bad = int((img.view("<u4") != ref.view("<u4")).any(axis=2).sum())
    print(name, W, H, spp, "kernel_ms %.2f" % st["kernel_ms"], "differing pixels", bad, flush=True)
    if bad: sys.exit(1)
