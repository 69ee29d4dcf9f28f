#!/usr/bin/env python3
"""bench.py -- headline benchmark of the render call (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W        (N > 1: starts its own N rank processes, see self_launch)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one full render of the workload through the C ABI (HIP persistent path-trace
kernel + chunk combine, output left in HBM) plus, for N > 1, the single RCCL gather of the
framebuffer to rank 0.  Workload = the configuration the metric is quoted on: the reference's
bunny.ply (69 451 triangles) in a closed room, 1920x1080, 1024 spp, rr 0.8 -- one frame,
sharded by 8x8-pixel blocks over the N GPUs (total work fixed => "strong" scaling).

Prints ONE JSON line on rank 0 with the contract fields plus
  roofline     : what bounds the kernel is VALU issue at partial lane utilisation (and the memory
                 latency behind it), not HBM bandwidth, so the headline figure is
                   bound "valu": achieved = useful vector lane-operations per second
                                 = SQ_INSTS_VALU x 64 x lanes_active / kernel time  (PMC stamp of THIS
                                 kernel source and workload, profiles/*pmc_stamp*.json; kernel time
                                 live from HIP events), peak = 78.6 T lane-op/s (157.3 TFLOP/s f32
                                 vector / 2), frac = achieved / peak.
                 The memory side stays in `hbm`: SURVEY 8d's algorithmic bytes from exact device
                 counters WITHOUT the analytic prologue's records (wave-uniform reads served from
                 the LDS tables, not memory traffic) against the 8 TB/s peak (`frac`), the PMC-measured
                 HBM bytes/s (`measured_gbs`; where the ray exchange runs about 8 % of the peak, mostly
                 the parking traffic of the stashes; under the plain loop of the headline 0.5 %), and 8d's figure as written (`contract_frac`,
                 saturated: kept for continuity only).  Without a stamp that matches the kernel
                 sources and the workload the line falls back to bound "hbm" on the divergent bytes
                 and says so.
  cpu_baseline : the reference's own code (oracle/_ref, kind "reference") or the oracle
                 (kind "port") timed on this host's cores on a bounded sample; N=1 only.

  --policy tile32 renders with main()'s own schedule instead (macos_main.mm:602-662: 1024 tiles,
  one serial RNG stream each = 1024 GPU lanes): what the reference's caller costs on a GPU.
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec peak
VALU_PEAK_TLANEOPS = 78.65  # same guide: 157.3 TFLOP/s f32 vector = 256 CUs x 128 lanes x 2.4 GHz x 2 flop -> lane-operations / 2


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--scene", default="c3_bunny_room")
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--spp", type=int, default=1024)
    ap.add_argument("--chunk", type=int, default=64)
    ap.add_argument("--seed", type=int, default=12345)
    ap.add_argument("--policy", default="chunk", choices=["chunk", "pixel", "tile32"],
                    help="seeding policy = the caller side of the reference call (include/ort.h); tile32 is main()'s schedule")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-spp", type=int, default=96)
    ap.add_argument("--strict-gather", action="store_true",
                    help="N > 1: fail instead of moving the blocks with torch.distributed when the C++ RCCL gather cannot be used")
    ap.add_argument("--rendezvous-only", action="store_true",
                    help="start the ranks, meet over the process group (gloo on a machine without a GPU), print one line, exit: "
                         "exercises the launch path without rendering")
    return ap.parse_args()


def self_launch(args):
    """`python bench.py --gpus N` started as ONE plain process: start N rank processes of this script (one per GPU,
    RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set as torch.distributed.run would set them), relay rank 0's JSON line,
    fail if any rank fails.  The parent never touches HIP or torch -- the ranks are CHILD processes started before
    anything here could have initialised the GPU, never an exec of this process.  Counterpart in the reference:
    main() starting its own worker threads (macos_main.mm:565-598)."""
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    rc = 0
    pending = set(range(args.gpus))
    while pending and rc == 0:
        for r in sorted(pending):
            code = procs[r].poll()
            if code is not None:
                pending.discard(r)
                if code != 0:
                    rc = code if code > 0 else 1
                    sys.stderr.write("bench.py: rank %d exited with code %d\n" % (r, code))
        if pending and rc == 0:
            time.sleep(0.05)
    if rc != 0:  # a dead rank leaves the others waiting in a collective: end exactly the processes started here
        for r in pending:
            procs[r].terminate()
        for r in pending:
            try:
                procs[r].wait(timeout=10)
            except subprocess.TimeoutExpired:
                procs[r].kill()
    # ONE JSON line on stdout: anything else rank 0's libraries wrote there (gloo prints its connection banner to stdout)
    # goes to stderr
    lines = procs[0].stdout.read().decode().splitlines()
    json_lines = [ln for ln in lines if ln.startswith("{")]
    for ln in lines:
        if not ln.startswith("{"):
            sys.stderr.write(ln + "\n")
    if rc == 0 and len(json_lines) != 1:
        sys.stderr.write("bench.py: rank 0 printed %d JSON lines, expected one\n" % len(json_lines))
        rc = 1
    for ln in json_lines:
        sys.stdout.write(ln + "\n")
    sys.stdout.flush()
    sys.exit(rc)


def cpu_baseline(args, scene_path):
    """The reference's own hot path (compiled from /root/reference by oracle/Makefile; the
    binary travels, the sources do not) on every host core, on a bounded sample of the same
    workload: the central 960x540 pixels of the 1920x1080 frame at a reduced spp, per-pixel
    seeds.  One process per core, each a horizontal band."""
    cores = os.cpu_count() or 1
    try:
        cores = len(os.sched_getaffinity(0))
    except Exception:
        pass
    # a 1-GPU box's CPU share is 16 cores (each reference process also reserves ~0.5 GB of arenas)
    cores = max(1, min(cores, int(os.environ.get("ORT_CPU_BASELINE_CORES", "16"))))
    W, H = args.width, args.height
    x0, x1 = W // 4, W - W // 4
    y0, y1 = H // 4, H - H // 4
    spp = args.cpu_sample_spp
    ref_bin = os.path.join(ROOT, "oracle", "_ref", "ref_glibc")
    sample = "central %dx%d px of the %dx%d frame, %d spp, per-pixel seeds" % (x1 - x0, y1 - y0, W, H, spp)
    # the reference build cannot hold the 1M-triangle mesh (its fixed arenas overflow): port baseline there
    if os.path.exists(ref_bin) and not args.scene.startswith("c5_heightfield_"):
        bands = [(y0 + (y1 - y0) * k // cores, y0 + (y1 - y0) * (k + 1) // cores) for k in range(cores)]
        procs = []
        for k, (a, b) in enumerate(bands):
            if b <= a:
                continue
            out = "/tmp/ort_cpu_baseline_%d_%d.f32" % (os.getpid(), k)
            cmd = [ref_bin, "render", scene_path, os.path.join(ROOT, "data") + "/", str(W), str(H), str(spp),
                   str(args.seed), "pixel", out, "1", str(x0), str(a), str(x1), str(b)]
            procs.append((subprocess.Popen(cmd, stdout=subprocess.PIPE), out))
        paths, slowest = 0.0, 0.0
        for p, out in procs:
            text = p.communicate()[0].decode().strip().splitlines()[-1]
            js = json.loads(text)
            paths += js["paths"]
            slowest = max(slowest, js["seconds"])
            try:
                os.remove(out)
            except OSError:
                pass
        return {"value": paths / slowest / 1e6, "unit": "Mpaths/s", "cores": len(procs), "kind": "reference",
                "sample": sample + "; reference ray.cpp built with clang++ -O2 -ffp-contract=off + glibc libm, "
                "one process per core (horizontal bands), render time only"}
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib
    from offline_raytracer_amd import api
    scene = api.Scene.load_scn(scene_path).commit()
    osc = oracle_lib.OracleScene(scene.flatten(W, H))
    if args.scene.startswith("c5_heightfield_"):  # the reference's octree needs ~5000 triangle tests per ray here
        x0, x1, y0, y1, spp = W // 2 - 64, W // 2 + 64, H // 2 - 32, H // 2 + 32, 2
        sample = "central %dx%d px of the %dx%d frame, %d spp, per-pixel seeds" % (x1 - x0, y1 - y0, W, H, spp)
    _, st = osc.render(W, H, spp, args.seed, "pixel", rect=(x0, y0, x1, y1), threads=cores)
    note = ("the reference build cannot hold this mesh (its fixed arenas overflow): the oracle walks the reference's octree instead"
            if args.scene.startswith("c5_heightfield_") else
            "oracle/_ref/ref_glibc (the reference's own sources compiled by `make -C oracle ref`, git-ignored, needs /root/reference) is "
            "absent on this machine: timed the plain-C restatement instead")
    return {"value": st["paths"] / st["seconds"] / 1e6, "unit": "Mpaths/s", "cores": cores, "kind": "port",
            "sample": sample + "; oracle/liboracle.so (plain-C restatement), pthreads", "note": note}


def kernel_source_hash():
    """SHA-256 over the sources the HIP kernels are built from: ties a PMC stamp to the code it was measured on."""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "offline_raytracer_amd", "csrc")
    for name in ("ort_lane.h", "ort_kernels.hip", "ort_kernels_w5.hip", "ort_device.h", "ort_detmath.h", "ort_scene.h"):
        with open(os.path.join(d, name), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def pmc_stamp(workload_key):
    """The PMC stamp of this workload (profiles/*pmc_stamp*.json, written by tools/make_pmc_stamp.py from a
    rocprofv3 --pmc run of the same render): returned only if it was measured on the kernel sources that are in
    the tree now."""
    import glob
    want = kernel_source_hash()
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*pmc_stamp*.json")), reverse=True):
        try:
            st = json.load(open(path))
        except Exception:
            continue
        if st.get("kernel_hash") == want and st.get("workload_key") == workload_key:
            st["file"] = os.path.relpath(path, ROOT)
            return st
    return None


def main():
    args = parse_args()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    launch_mode = ("torch.distributed.run" if "TORCHELASTIC_RUN_ID" in os.environ else
                   ("self-launched rank processes" if "WORLD_SIZE" in os.environ else "single process"))
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        self_launch(args)  # a plain `python bench.py --gpus N`: this process only starts and watches the ranks
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        args.gpus = world

    import torch
    import torch.distributed as dist

    if os.environ.get("ORT_BENCH_TEST_FAIL_RANK") == str(rank):
        sys.exit(3)  # tests/test_bench_launch.py: a rank that dies must fail the whole launch, promptly
    if args.rendezvous_only:
        # the launch path alone: every rank meets the others over the process group (what the bench's barrier, max-over-
        # ranks timing and id broadcast ride on) and rank 0 prints one line.  gloo where there is no GPU (CPU tests).
        on_gpu = torch.cuda.is_available() and os.environ.get("ORT_BENCH_SHARE_GPU") != "1" and torch.cuda.device_count() >= world
        if world > 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            if on_gpu:
                torch.cuda.set_device(local_rank)
                dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
            else:
                dist.init_process_group("gloo", rank=rank, world_size=world)
            t = torch.tensor([float(rank + 1)], device=torch.device("cuda", local_rank) if on_gpu else torch.device("cpu"))
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
            total = float(t.item())
            dist.barrier()
            dist.destroy_process_group()
        else:
            total = 1.0
        if rank == 0:
            print(json.dumps({"rendezvous": "ok", "world": world, "backend": "nccl" if (on_gpu and world > 1) else "gloo",
                              "rank_sum": total, "expected": world * (world + 1) / 2.0}), flush=True)
        return

    from offline_raytracer_amd import api, dist as odist

    if not torch.cuda.is_available() or api.device_count() < 1:
        sys.exit("bench.py needs a HIP device: the render path has no CPU fallback")
    # one process per GPU.  ORT_BENCH_SHARE_GPU=1 is a rehearsal mode for 1-GPU boxes: all ranks use
    # device 0 and the gather goes through gloo on host copies (RCCL refuses two ranks on one device);
    # the numbers of such a run mean nothing, it only exercises the N > 1 code path.
    share_gpu = os.environ.get("ORT_BENCH_SHARE_GPU") == "1"
    device_index = 0 if share_gpu else local_rank
    torch.cuda.set_device(device_index)
    dev = torch.device("cuda", device_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if share_gpu:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    if args.scene.startswith("c5_heightfield_"):
        # BASELINE.json configs[4]: the synthetic mesh is generated, not stored (tools/make_heightfield.py)
        import tempfile
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import make_heightfield
        scene_path, _, _ = make_heightfield.write_scene(int(args.scene.rsplit("_", 1)[1]), tempfile.mkdtemp(prefix="c5_"))
    else:
        scene_path = os.path.join(ROOT, "data", args.scene + ".scn")
    scene = api.Scene.load_scn(scene_path).commit().upload(device_index)
    W, H = args.width, args.height
    if args.policy == "tile32" and world > 1:
        sys.exit("bench.py --policy tile32 is a one-GPU line: main()'s 1024 serial tiles do not shard by blocks")
    stream = torch.cuda.current_stream().cuda_stream
    chunk = args.chunk if args.policy == "chunk" else 0
    # The step: this rank renders ITS 8x8 blocks into a packed buffer (1/N of a frame; the CHUNK partial sums use the
    # same layout) and ONE collective assembles the frame on rank 0 -- ort_gather_framebuffer: grouped ncclSend/ncclRecv
    # over RCCL + an un-permute kernel, all inside libort.so.  If RCCL cannot be brought up from C++ (or in the
    # one-GPU rehearsal mode) the blocks travel through torch.distributed instead; `gather` in the line says which.
    sharded, gather_impl = None, "none (1 GPU): un-permute kernel only"
    if args.policy != "tile32":
        try:
            if share_gpu and world > 1:
                raise RuntimeError("rehearsal mode: RCCL refuses two ranks on one device")
            sharded = odist.ShardedRenderer(scene, W, H, rank, world, device_index)
            if world > 1:
                gather_impl = "ort_gather_framebuffer (C++: RCCL ncclSend/ncclRecv group + un-permute kernel)"
        except Exception as e:  # noqa: BLE001 -- any failure to bring RCCL up must not lose the measurement
            sharded = None
            gather_impl = "torch.distributed.gather of packed blocks (C++ RCCL path unavailable: %s)" % str(e)[:120]
        if world > 1:  # all ranks must take the same path
            ok = torch.tensor([1 if sharded is not None else 0], device=torch.device("cpu") if share_gpu else dev)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            if int(ok.item()) == 0:
                sharded = None
            if sharded is None and not share_gpu:
                if args.strict_gather:
                    sys.exit("bench.py --strict-gather: the C++ RCCL gather is unavailable (%s)" % gather_impl)
                if rank == 0:
                    sys.stderr.write("bench.py: WARNING: %s\n" % gather_impl)
    fb = sharded.full if (sharded is not None and rank == 0) else torch.zeros((H, W, 3), dtype=torch.float32, device=dev)
    if sharded is not None:
        params = sharded.params(args.spp, args.seed, args.policy, chunk=chunk)
    else:
        params = api.Scene.params(W, H, args.spp, args.seed, args.policy, chunk=chunk, shard=(rank, world))
    if sharded is not None and world > 1:
        # probe: one small packed render + the C++ gather, before anything is timed.  If it fails on any rank (an RCCL
        # error the communicator's creation did not show), every rank switches to the torch.distributed gather together.
        probe_err = ""
        try:
            sharded.render(sharded.params(1, args.seed, "pixel"), stream=stream, want_stats=True)
            sharded.gather(stream=stream)
            torch.cuda.synchronize()
        except Exception as e:  # noqa: BLE001
            probe_err = str(e)[:120] or "error"
        ok = torch.tensor([0 if probe_err else 1], device=dev)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if int(ok.item()) == 0:
            sharded = None
            gather_impl = "torch.distributed.gather of packed blocks (C++ RCCL gather failed in the probe step: %s)" % (probe_err or "on another rank")
            if args.strict_gather:
                sys.exit("bench.py --strict-gather: %s" % gather_impl)
            if rank == 0:
                sys.stderr.write("bench.py: WARNING: %s\n" % gather_impl)
            fb = torch.zeros((H, W, 3), dtype=torch.float32, device=dev)
            params = api.Scene.params(W, H, args.spp, args.seed, args.policy, chunk=chunk, shard=(rank, world))

    def step(want_stats=False):
        if sharded is not None:
            st = sharded.render(params, stream=stream, want_stats=want_stats)
            sharded.gather(stream=stream)
            return st
        st = scene.render_device(fb.data_ptr(), params, stream=stream, want_stats=want_stats)
        if world > 1:
            if share_gpu:
                torch.cuda.synchronize()
                odist.gather_framebuffer(fb.cpu(), W, H, rank, world)
            else:
                odist.gather_framebuffer(fb, W, H, rank, world)
        return st

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    if args.warmup == 0:
        step()  # the first call allocates the workspace (partial planes, staging): never inside the timed region
    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    kernel_ms = []
    for _ in range(args.steps):
        st = step(want_stats=True)  # HIP events around the kernel, on the launch stream
        kernel_ms.append(st["kernel_ms"])
    fence()
    elapsed = time.perf_counter() - t0
    t = torch.tensor([elapsed], dtype=torch.float64, device=torch.device("cpu") if share_gpu else dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    ms_per_step = float(t.item()) / args.steps * 1e3
    paths_per_step = float(W) * H * args.spp
    value = paths_per_step / (ms_per_step * 1e-3) / 1e6

    # algorithmic bytes per path from exact device counters (untimed counters build of the same
    # kernel, same scene/seed/policy at reduced spp: the per-path averages are spp-independent)
    cspp = args.chunk if args.policy == "chunk" else min(args.spp, 4)
    cparams = api.Scene.params(W, H, cspp, args.seed, args.policy, chunk=chunk, counters=True, shard=(rank, world))
    cfb = torch.zeros((H, W, 3), dtype=torch.float32, device=dev)
    cst = scene.render_device(cfb.data_ptr(), cparams, stream=stream, want_stats=True)
    R = cst["rays"] / max(1, cst["paths"])
    Vn = cst["node_tests"] / max(1, cst["rays"])
    Vt = cst["tri_tests"] / max(1, cst["rays"])
    Vp = cst["analytic_tests"] / max(1, cst["rays"])
    bytes_per_path = R * (Vn * 32.0 + Vt * 36.0 + Vp * 32.0) + 12.0 / args.spp
    # the analytic prologue's shape records are read once per workgroup into LDS and broadcast: they are not memory
    # traffic in any physical sense.  Counting only divergent reads (tree nodes, triangles, analytic shapes IN the tree):
    pro = scene.tree_info().get("prologue_prims", 0)
    bytes_per_path_divergent = R * (Vn * 32.0 + Vt * 36.0 + max(0.0, Vp - pro) * 32.0) + 12.0 / args.spp
    my_paths = paths_per_step / world  # per launch on this rank (blocks are dealt round-robin)
    k_ms = sum(kernel_ms) / len(kernel_ms)
    achieved = bytes_per_path * my_paths / (k_ms * 1e-3) / 1e9
    # measured figures come from a separate rocprofv3 --pmc run of THIS step (the profiler cannot be attached from
    # inside); they are quoted only while the stamp matches the kernel sources in the tree and the workload
    workload_key = "%s %dx%d %dspp %s%d x%d" % (args.scene, W, H, args.spp, args.policy, args.chunk if args.policy == "chunk" else 0, world)
    stamp = pmc_stamp(workload_key)
    traffic = stamp.get("hbm_bytes_per_launch") if stamp else None
    achieved_divergent = bytes_per_path_divergent * my_paths / (k_ms * 1e-3) / 1e9

    if rank == 0:
        k_s = k_ms * 1e-3
        hbm = {"achieved": achieved_divergent, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved_divergent / HBM_PEAK_GBS,
               "bytes_per_path": bytes_per_path_divergent,
               "what": "SURVEY 8d's algorithmic bytes from exact device counters, without the analytic prologue's records (wave-uniform reads "
                       "served from the LDS tables), per launch / kernel time",
               "measured_gbs": (traffic / k_s / 1e9) if traffic else None,
               "measured_frac": (traffic / k_s / 1e9 / HBM_PEAK_GBS) if traffic else None,
               "l2_hit_rate": stamp.get("l2_hit_rate") if stamp else None,
               "contract_bytes_per_path": bytes_per_path, "contract_achieved": achieved, "contract_frac": achieved / HBM_PEAK_GBS,
               "contract_note": "8d's formula as written prices the prologue's LDS-served records as HBM bytes: saturated, kept for continuity only"}
        if stamp and stamp.get("counters", {}).get("SQ_INSTS_VALU"):
            # useful vector lane-operations per second: wave-level VALU instructions of one launch (PMC) x 64 lanes x the
            # fraction of lanes that were active in them, over the kernel time measured live
            valu_insts = float(stamp["counters"]["SQ_INSTS_VALU"])
            useful = valu_insts * 64.0 * float(stamp["lanes_active"]) / k_s / 1e12
            roofline = {"bound": "valu", "achieved": useful, "peak": VALU_PEAK_TLANEOPS, "unit": "Tlane-op/s", "frac": useful / VALU_PEAK_TLANEOPS,
                        "traffic": traffic, "valu_insts_per_launch": valu_insts, "lanes_active": stamp.get("lanes_active"),
                        "valu_issue_frac": valu_insts / k_s / (VALU_PEAK_TLANEOPS * 1e12 / 64.0),
                        "valu_issue_frac_at_stamp": stamp.get("valu_issue_frac"),
                        "wave_cycles_waiting_on_memory": stamp.get("wave_cycles_waiting_on_memory"),
                        "limiter": "VALU issue at partial lane utilisation (divergence) with memory latency behind it; not HBM bandwidth"}
        else:
            roofline = {"bound": "hbm", "achieved": achieved_divergent, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": achieved_divergent / HBM_PEAK_GBS, "traffic": None, "lanes_active": None, "valu_issue_frac": None,
                        "limiter": "no PMC stamp matches these kernel sources and this workload (profiles/*pmc_stamp*.json): "
                                   "falling back to the divergent algorithmic bytes against the HBM peak"}
        roofline.update({
            "hbm": hbm, "frac_divergent": achieved_divergent / HBM_PEAK_GBS,
            "pmc_stamp": ({"file": stamp.get("file"), "kernel_hash": stamp["kernel_hash"], "git_commit": stamp.get("git_commit"),
                           "source": stamp.get("source"), "workload_key": stamp.get("workload_key")} if stamp else None),
            "kernel_hash": kernel_source_hash(), "workload_key": workload_key,
            "kernel": "pt_persistent", "kernel_ms": k_ms,
            "rays_per_path": R, "node_tests_per_ray": Vn, "tri_tests_per_ray": Vt,
            "analytic_tests_per_ray": Vp, "prologue_tests_per_ray": pro, "fallback_rays": cst["fallback_rays"]})
        line = {
            "metric": "Mpaths/s", "value": value, "unit": "Mpaths/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f32",
            "data": "synthetic: build-authored closed-room .scn around %s, fixed seed %d"
                    % ("a generated height-field mesh" if args.scene.startswith("c5_") else "the reference's mesh files", args.seed),
            "config": {"workload": "%s.scn (%d triangles) %dx%d %dspp rr0.8, %s, "
                                   "one frame sharded in 8x8 blocks over %d GPU(s)" % (args.scene, scene.info().triangle_count, W, H, args.spp,
                                   ("CHUNK policy chunk=%d" % args.chunk) if args.policy == "chunk" else
                                   ("TILE32 policy: main()'s 1024 tiles, one serial stream each" if args.policy == "tile32" else "PIXEL policy"), world),
                       "width": W, "height": H, "spp": args.spp, "paths_per_step": paths_per_step, "gather": gather_impl,
                       "gather_fallback": bool(world > 1 and sharded is None),
                       "launch": launch_mode, "workspace_bytes_per_rank": api.workspace_bytes(params)},
            "roofline": roofline,
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(args, scene_path)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
