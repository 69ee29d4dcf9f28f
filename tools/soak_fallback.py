"""Developer script: the pooled reference-order fallback under heavy contention -- full-size frames with a
forced re-cast rate of 1/16 .. 1/256 of the rays must equal the normal render bit for bit.
usage: python3 tools/soak_fallback.py"""
import os, sys
os.environ.setdefault("ORT_KNOBS_LIVE", "1")  # this script flips knobs between renders of one uploaded scene
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from offline_raytracer_amd import api
bad = 0
for name, w, h, spp in (("c3_bunny_room", 1920, 1080, 8), ("testscene", 1280, 720, 4), ("glass_room", 1920, 1080, 8)):
    scene = api.Scene.load_scn(os.path.join(ROOT, "data", name + ".scn")).commit().upload(0)
    os.environ.pop("ORT_DEBUG_FORCE_FALLBACK", None)
    ref, st0 = scene.render(w, h, spp, 99, "chunk", chunk=4, counters=True)
    for mask in ("0xff", "0xf"):
        os.environ["ORT_DEBUG_FORCE_FALLBACK"] = mask
        img, st = scene.render(w, h, spp, 99, "chunk", chunk=4, counters=True)
        diff = int((img.view("<u4") != ref.view("<u4")).any(axis=2).sum())
        bad += diff != 0
        print("%-14s %dx%d %dspp mask %-5s rays %d re-cast %d (normally %d) kernel %.0f ms: %s" %
              (name, w, h, spp, mask, st["rays"], st["fallback_rays"], st0["fallback_rays"], st["kernel_ms"], "ok" if diff == 0 else "DIFF %d px" % diff), flush=True)
    scene.close()
sys.exit(1 if bad else 0)
