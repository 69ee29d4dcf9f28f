import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from offline_raytracer_amd import api
scene = api.Scene.load_scn(os.path.join(ROOT, "data", (sys.argv[1] if len(sys.argv) > 1 else "c3_bunny_room") + ".scn")).commit().upload(0)
img, st = scene.render(960, 540, 16, 12345, "chunk", chunk=16, counters=True)
print(os.environ.get("ORT_LIB", "default")[-14:], {k: (round(v / st["rays"], 3) if k.endswith("tests") else v) for k, v in st.items()}, flush=True)
