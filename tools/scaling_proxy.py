"""One-GPU proxy for the N-GPU scaling of the headline frame (developer tool; run on the GPU box).

Every shard (r, N) of the frame -- the 8x8 blocks rank r of N would render, packed -- is rendered alone on this GPU
and timed with HIP events (the path-trace kernel + the chunk combine, as bench.py's step minus the gather).  An N-GPU
step lasts as long as its slowest rank, so  predicted speed-up = full_ms / max_r shard_ms  (the gather of 24.9 MB / N per
rank over xGMI, ~0.1 ms, and launch skew are not in it).  Shards do not share a bit of state, so what this cannot see is
only the transfer itself.

usage: python3 tools/scaling_proxy.py [scene] [W] [H] [spp] [chunk] [out.json]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402  (device buffers)
from offline_raytracer_amd import api  # noqa: E402

a = sys.argv[1:]
name = a[0] if len(a) > 0 else "c3_bunny_room"
W = int(a[1]) if len(a) > 1 else 1920
H = int(a[2]) if len(a) > 2 else 1080
spp = int(a[3]) if len(a) > 3 else 1024
chunk = int(a[4]) if len(a) > 4 else 64
out = a[5] if len(a) > 5 else None
reps = int(os.environ.get("PROXY_REPS", "2"))
worlds = [int(x) for x in os.environ.get("PROXY_WORLDS", "1,2,4,8").split(",")]

scene = api.Scene.load_scn(os.path.join(ROOT, "data", name + ".scn")).commit().upload(0)
stream = torch.cuda.current_stream().cuda_stream
res = {"workload": "%s %dx%d %dspp chunk%d" % (name, W, H, spp, chunk), "reps": reps, "worlds": {}}
full_ms = None
for N in worlds:
    per_rank = []
    for r in range(N):
        n = api.shard_block_count(W, H, r, N)
        packed = torch.zeros((max(1, n), 64, 3), dtype=torch.float32, device="cuda")
        p = api.Scene.params(W, H, spp, 12345, "chunk", chunk=chunk, shard=(r, N), packed=True)
        best = None
        for _ in range(reps):
            st = scene.render_device(packed.data_ptr(), p, stream=stream, want_stats=True)
            best = st["kernel_ms"] if best is None else min(best, st["kernel_ms"])
        per_rank.append(best)
        del packed
    slow = max(per_rank)
    if N == 1:
        full_ms = slow
    res["worlds"][str(N)] = {"shard_ms": per_rank, "max_ms": slow, "mean_ms": sum(per_rank) / N,
                             "predicted_speedup": (full_ms / slow) if full_ms else None,
                             "imbalance": slow / (sum(per_rank) / N),
                             "fixed_ms_estimate": (slow - full_ms / N) if full_ms else None,
                             "predicted_mpaths": W * H * spp / slow / 1e3}
    print("N=%d max %.2f ms mean %.2f ms -> x%.2f (%.0f Mpaths/s)" % (N, slow, sum(per_rank) / N, (full_ms / slow) if full_ms else 1.0,
                                                                       W * H * spp / slow / 1e3), flush=True)
res["env"] = {k: v for k, v in os.environ.items() if k.startswith("ORT_") and k != "ORT_LIB"}
if out:
    json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res))
