"""The multi-GPU side of the render call on ONE GPU: packed per-shard framebuffers (ORT_RENDER_PACKED), the
un-permute kernel, the world = 1 gather through the C ABI, the per-rank workspace, and bin/ort_render --gpus N with
all shards on device 0.  What cannot run here is the RCCL transfer itself (RCCL refuses two ranks on one device):
ranks > 1 on real GPUs are the driver's scaling run; the CPU side of the collective is tests/test_distributed_cpu.py."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from conftest import DATA, ROOT, assert_bits_equal

pytestmark = pytest.mark.gpu


def _torch():
    import torch
    assert torch.cuda.is_available()
    return torch


@pytest.mark.parametrize("w,h,world", [(200, 120, 1), (203, 117, 3), (1920, 1080, 8)])
def test_packed_shards_unpermute_to_the_whole_frame(api, gpu_scene, w, h, world):
    """every shard renders into its packed buffer; un-permuting them all (device kernel) gives the one-GPU image"""
    torch = _torch()
    scene = gpu_scene("c3_bunny_room")
    spp, chunk, seed = 8, 4, 4242
    whole, _ = scene.render(w, h, spp, seed, "chunk", chunk=chunk)
    full = torch.zeros((h, w, 3), dtype=torch.float32, device="cuda")
    total = 0
    for r in range(world):
        n = api.shard_block_count(w, h, r, world)
        total += n
        packed = torch.full((max(1, n), 64, 3), -1.0, dtype=torch.float32, device="cuda")
        p = api.Scene.params(w, h, spp, seed, "chunk", chunk=chunk, shard=(r, world), packed=True)
        assert api.workspace_bytes(p) == (spp // chunk) * n * 768  # partial sums: this shard's blocks only
        scene.render_device(packed.data_ptr(), p, want_stats=True)
        api.unpack_blocks_device(packed.data_ptr(), w, h, r, world, full.data_ptr())
        # the packed buffer equals the host-side packing of the whole image
        torch.cuda.synchronize()
        want = api.pack_blocks_host(whole, r, world)
        got = packed.cpu().numpy()[:n]
        inside = np.ones((n, 64), bool)  # pixels of ragged edge blocks that fall outside the image are never written
        gw = (w + 7) // 8
        for k in range(n):
            blk = r + k * world
            xs = (blk % gw) * 8 + (np.arange(64) & 7)
            ys = (blk // gw) * 8 + (np.arange(64) >> 3)
            inside[k] = (xs < w) & (ys < h)
        assert_bits_equal(got[inside], want[inside], "packed blocks of shard %d/%d" % (r, world))
    torch.cuda.synchronize()
    assert total == ((w + 7) // 8) * ((h + 7) // 8)
    assert_bits_equal(full.cpu().numpy(), whole, "un-permuted union of %d shards" % world)


def test_sharded_renderer_world_one_goes_through_the_c_gather(api, gpu_scene):
    """offline_raytracer_amd.dist.ShardedRenderer (what bench.py steps): packed render + ort_gather_framebuffer"""
    torch = _torch()
    from offline_raytracer_amd import dist as odist
    scene = gpu_scene("c2_analytic")
    w, h, spp, seed = 160, 90, 6, 7
    sr = odist.ShardedRenderer(scene, w, h, 0, 1, 0)
    stream = torch.cuda.current_stream().cuda_stream
    st = sr.render(sr.params(spp, seed, "pixel"), stream=stream, want_stats=True)
    full = sr.gather(stream=stream)
    torch.cuda.synchronize()
    ref, _ = scene.render(w, h, spp, seed, "pixel")
    assert st["kernel_ms"] > 0
    assert_bits_equal(full.cpu().numpy(), ref, "ShardedRenderer world=1")


def test_gather_through_a_real_rccl_communicator(api, gpu_scene, monkeypatch):
    """ORT_COMM_FORCE_RCCL: a world of one gets a real ncclComm (ncclGetUniqueId + ncclCommInitRank through the
    dlopen'd entry points) and the gather moves the packed blocks rank 0 -> rank 0 with the grouped ncclSend / ncclRecv
    of the N-rank path, on the caller's stream, before the un-permute kernel: everything of the collective that can run
    on one GPU"""
    torch = _torch()
    from offline_raytracer_amd import dist as odist
    monkeypatch.setenv("ORT_COMM_FORCE_RCCL", "1")
    scene = gpu_scene("c4_dwarf_room")
    w, h, spp, seed = 203, 117, 4, 3
    sr = odist.ShardedRenderer(scene, w, h, 0, 1, 0)
    stream = torch.cuda.current_stream().cuda_stream
    sr.render(sr.params(spp, seed, "chunk", chunk=2), stream=stream, want_stats=True)
    full = sr.gather(stream=stream)
    torch.cuda.synchronize()
    ref, _ = scene.render(w, h, spp, seed, "chunk", chunk=2)
    assert_bits_equal(full.cpu().numpy(), ref, "gather through RCCL, world = 1")
    sr.comm.close()


def test_packed_shards_with_the_ray_exchange(api, gpu_scene, monkeypatch):
    """what bench.py steps on the headline: packed output + CHUNK partial planes + the ray exchange, here forced on for
    two of three shards of a small frame; the union must be the plain one-GPU image"""
    torch = _torch()
    scene = gpu_scene("c3_bunny_room")
    w, h, spp, chunk, seed, world = 320, 200, 16, 4, 99, 3
    monkeypatch.setenv("ORT_EXCHANGE", "0")
    whole, _ = scene.render(w, h, spp, seed, "chunk", chunk=chunk)
    full = torch.zeros((h, w, 3), dtype=torch.float32, device="cuda")
    for r in range(world):
        monkeypatch.setenv("ORT_EXCHANGE", "1" if r != 1 else "0")
        n = api.shard_block_count(w, h, r, world)
        packed = torch.zeros((max(1, n), 64, 3), dtype=torch.float32, device="cuda")
        scene.render_device(packed.data_ptr(), api.Scene.params(w, h, spp, seed, "chunk", chunk=chunk, shard=(r, world), packed=True), want_stats=True)
        api.unpack_blocks_device(packed.data_ptr(), w, h, r, world, full.data_ptr())
        torch.cuda.synchronize()
    assert_bits_equal(full.cpu().numpy(), whole, "packed shards, exchange on")


def test_per_rank_workspace_at_the_stress_config(api):
    """BASELINE.json configs[4] on 8 GPUs: 3840x2160, 4096 spp in 64-sample jobs -- a rank keeps 1/8 of the partial
    planes (round 1: full frames, 6.4 GB per rank)"""
    full = api.workspace_bytes(api.Scene.params(3840, 2160, 4096, 1, "chunk", chunk=64))
    assert full == 64 * 3840 * 2160 * 12
    worst = max(api.workspace_bytes(api.Scene.params(3840, 2160, 4096, 1, "chunk", chunk=64, shard=(r, 8), packed=True)) for r in range(8))
    assert worst <= 0.8e9 + 1e6 and worst * 8 <= full + 8 * 64 * 768


def test_packed_needs_a_per_pixel_policy_and_the_full_rect(api, gpu_scene):
    scene = gpu_scene("c2_analytic")
    out = np.zeros((32, 32, 3), "<f4")
    for kw in (dict(policy="tile32"), dict(policy="pixel", rect=(0, 0, 16, 32))):
        p = api.Scene.params(32, 32, 2, 1, packed=True, **kw)
        with pytest.raises(api.OrtError):
            scene.render_device(out.ctypes.data, p)


def test_cli_gpus_flag_shards_and_gathers(api, gpu_scene, tmp_path):
    """bin/ort_render --gpus 3 (all three shards on device 0: ORT_CLI_SHARE_DEVICE) writes the file --gpus 1 writes"""
    cli = os.path.join(ROOT, "offline_raytracer_amd", "bin", "ort_render")
    base = [cli, "--scene", os.path.join(DATA, "c4_dwarf_room.scn"), "--width", "100", "--height", "60", "--spp", "8", "--chunk", "4", "--seed", "5"]
    one, many = str(tmp_path / "one.f32"), str(tmp_path / "many.f32")
    r = subprocess.run(base + ["--raw", one, "--out", str(tmp_path / "one.hdr")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    r = subprocess.run(base + ["--gpus", "3", "--raw", many, "--out", str(tmp_path / "many.hdr")], capture_output=True, text=True,
                       env=dict(os.environ, ORT_CLI_SHARE_DEVICE="1"))
    assert r.returncode == 0, r.stderr
    assert "on 3 GPU(s)" in r.stdout
    assert_bits_equal(np.fromfile(many, "<f4"), np.fromfile(one, "<f4"), "ort_render --gpus 3 vs --gpus 1")
    assert open(tmp_path / "many.hdr", "rb").read() == open(tmp_path / "one.hdr", "rb").read()
