"""`python bench.py --gpus N` as ONE plain process (how the driver starts the 1-GPU line, and possibly the scaling
run): the parent starts N rank processes itself -- children, before anything touches a GPU -- relays rank 0's single
JSON line and fails if a rank fails.  Here on CPU with --rendezvous-only: the ranks meet over gloo and do not render
(the render path has no CPU fallback).  Reference counterpart: main() starting its own workers, macos_main.mm:565-598."""
import json
import os
import subprocess
import sys
import time

from conftest import ROOT

BENCH = os.path.join(ROOT, "bench.py")


def _run(args, env=None, timeout=180):
    e = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "TORCHELASTIC_RUN_ID"):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, BENCH] + args, env=e, capture_output=True, text=True, timeout=timeout)


def test_plain_process_starts_its_own_ranks():
    r = _run(["--gpus", "2", "--rendezvous-only"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout  # ONE JSON line, whatever the ranks' libraries print
    js = json.loads(lines[0])
    assert js["rendezvous"] == "ok" and js["world"] == 2 and js["rank_sum"] == js["expected"] == 3.0


def test_three_ranks():
    r = _run(["--gpus", "3", "--rendezvous-only"])
    assert r.returncode == 0, r.stderr[-2000:]
    js = json.loads(r.stdout.strip().splitlines()[-1])
    assert js["world"] == 3 and js["rank_sum"] == 6.0


def test_a_dead_rank_fails_the_launch_promptly():
    t0 = time.time()
    r = _run(["--gpus", "2", "--rendezvous-only"], env={"ORT_BENCH_TEST_FAIL_RANK": "1"}, timeout=120)
    assert r.returncode != 0
    assert "rank 1 exited with code 3" in r.stderr
    assert time.time() - t0 < 60  # rank 0 is not left waiting in the rendezvous
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]


def test_torchrun_form_still_works():
    """the contract's N > 1 launch: python -m torch.distributed.run ... bench.py --gpus N"""
    e = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    port = 29600 + os.getpid() % 300
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), BENCH, "--gpus", "2", "--rendezvous-only"], env=e, capture_output=True, text=True, timeout=240)
    assert r.returncode == 0, r.stderr[-2000:]
    js = [json.loads(ln) for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(js) == 1 and js[0]["world"] == 2
