"""One-GPU measurement of what a second frame in flight is worth (developer tool; run on the GPU box).

A launch costs 8-11 ms beyond its share of the work: the chip drains at the end (jobs are serial 64-sample streams).  A
caller that renders a SEQUENCE of frames can hide that: the scene is uploaded twice (two handles = two sets of per-render
device state: job counter, partial planes, render view), frames alternate between the handles on two HIP streams, and the
next frame's workgroups take the slots the draining one frees.  Timed here: K frames one at a time on one handle against
K frames alternating over two, whole frame (N = 1) and one 8-way shard of it; wall clock around a device synchronise.
bench.py does NOT do this: its step is one frame at a time.

usage: python3 tools/pipeline_proxy.py [scene] [W] [H] [spp] [chunk] [frames]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from offline_raytracer_amd import api  # noqa: E402

a = sys.argv[1:]
name = a[0] if len(a) > 0 else "c3_bunny_room"
W = int(a[1]) if len(a) > 1 else 1920
H = int(a[2]) if len(a) > 2 else 1080
spp = int(a[3]) if len(a) > 3 else 1024
chunk = int(a[4]) if len(a) > 4 else 64
K = int(a[5]) if len(a) > 5 else 8

path = os.path.join(ROOT, "data", name + ".scn")
handles = [api.Scene.load_scn(path).commit().upload(0) for _ in range(2)]
streams = [torch.cuda.Stream() for _ in range(2)]
for N, r in ((1, 0), (8, 0), (8, 4)):
    n = api.shard_block_count(W, H, r, N)
    outs = [torch.zeros((max(1, n), 64, 3), dtype=torch.float32, device="cuda") for _ in range(2)]
    p = api.Scene.params(W, H, spp, 12345, "chunk", chunk=chunk, shard=(r, N), packed=True)
    res = {}
    for mode, nh in (("one frame at a time", 1), ("two frames in flight", 2)):
        for h in range(nh):  # warm-up: workspaces allocated, code loaded
            handles[h].render_device(outs[h].data_ptr(), p, stream=streams[h].cuda_stream, want_stats=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(K):
            h = k % nh
            handles[h].render_device(outs[h].data_ptr(), p, stream=streams[h].cuda_stream, want_stats=False)
        torch.cuda.synchronize()
        res[mode] = (time.perf_counter() - t0) * 1e3 / K
    same = bool((outs[0].view(torch.int32) == outs[1].view(torch.int32)).all())
    print("shard (%d,%d): %.2f ms per frame one at a time, %.2f ms with two in flight (x%.3f); the two handles' images bit-equal: %s"
          % (r, N, res["one frame at a time"], res["two frames in flight"], res["one frame at a time"] / res["two frames in flight"], same), flush=True)
