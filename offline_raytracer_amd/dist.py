"""Multi-GPU plumbing for the render call -- a thin caller of the C++ side (csrc/ort_comm.cpp).

One process per GPU.  The path shards without any data-path collective: the scene (a few MB) is
replicated, every rank renders the 8x8-pixel blocks with block_id % world == rank straight into a
PACKED buffer [local block][pixel in block][rgb] (ORT_RENDER_PACKED; cost per pixel varies ~10x
across the image, so interleaving balances the ranks), and the seeding policies are per pixel, so
the union of the shards is bit-identical to a one-GPU render.  ONE collective per frame brings the
packed blocks to rank 0:

* on GPUs: `ShardedRenderer.gather` -> `ort_gather_framebuffer`: grouped ncclSend / ncclRecv over
  RCCL (each peer->root transfer rides its own xGMI link) + an un-permute kernel on rank 0.  The
  ncclUniqueId travels through torch.distributed (any backend); nothing else of torch is involved.
* on CPU (tests, gloo): `gather_framebuffer` packs with `ort_pack_blocks_host`, moves the packed
  blocks with torch.distributed.gather and un-permutes with `ort_unpack_blocks_host`.

The reference has no counterpart (one shared-memory process, code/macos_main.mm:565-671).
"""
import numpy as np
import torch
import torch.distributed as dist

from . import api

BLOCK = 8  # must match the kernel's implicit job space (ort_lane.h: 8x8 blocks)


def block_grid(width, height):
    return (width + BLOCK - 1) // BLOCK, (height + BLOCK - 1) // BLOCK


def my_block_ids(width, height, rank, world, device=None):
    bw, bh = block_grid(width, height)
    return torch.arange(rank, bw * bh, world, device=device)


def pack_blocks(image, rank, world):
    """image: [H, W, 3] float32 CPU tensor -> [n_my_blocks, 8*8*3] (this rank's blocks, row-major block order)."""
    packed = api.pack_blocks_host(image.detach().cpu().numpy(), rank, world)
    return torch.from_numpy(packed.reshape(packed.shape[0], BLOCK * BLOCK * 3))


def unpack_blocks(packed_per_rank, width, height):
    """inverse of pack_blocks over all ranks -> [H, W, 3]."""
    world = len(packed_per_rank)
    out = np.zeros((height, width, 3), "<f4")
    for r, p in enumerate(packed_per_rank):
        n = api.shard_block_count(width, height, r, world)
        api.unpack_blocks_host(p.detach().cpu().numpy()[:n], width, height, r, world, out=out)
    return torch.from_numpy(out)


def _pack_blocks_torch(image, rank, world):
    """the same packing with torch ops, for tensors that live on a device (fallback path only)"""
    height, width, _ = image.shape
    bw, bh = block_grid(width, height)
    padded = image
    if bw * BLOCK != width or bh * BLOCK != height:
        padded = torch.zeros((bh * BLOCK, bw * BLOCK, 3), dtype=image.dtype, device=image.device)
        padded[:height, :width] = image
    blocks = padded.view(bh, BLOCK, bw, BLOCK, 3).permute(0, 2, 1, 3, 4).reshape(bw * bh, BLOCK * BLOCK * 3)
    return blocks.index_select(0, my_block_ids(width, height, rank, world, image.device)).contiguous()


def _unpack_blocks_torch(packed_per_rank, width, height):
    world = len(packed_per_rank)
    bw, bh = block_grid(width, height)
    ref = packed_per_rank[0]
    blocks = torch.zeros((bw * bh, BLOCK * BLOCK * 3), dtype=ref.dtype, device=ref.device)
    for r, p in enumerate(packed_per_rank):
        ids = my_block_ids(width, height, r, world, ref.device)
        blocks.index_copy_(0, ids, p[: len(ids)])
    img = blocks.view(bh, bw, BLOCK, BLOCK, 3).permute(0, 2, 1, 3, 4).reshape(bh * BLOCK, bw * BLOCK, 3)
    return img[:height, :width].contiguous()


def gather_framebuffer(local_image, width, height, rank, world, group=None):
    """torch.distributed form of the collective (CPU / gloo in the tests; on device tensors the fallback of
    bench.py when RCCL cannot be brought up from C++).  Every rank passes a full-size framebuffer (only its own
    blocks are meaningful); rank 0 gets the assembled [H, W, 3] image, others None."""
    if world == 1:
        return local_image
    on_device = local_image.is_cuda
    bw, bh = block_grid(width, height)
    max_blocks = (bw * bh + world - 1) // world
    packed = _pack_blocks_torch(local_image, rank, world) if on_device else pack_blocks(local_image, rank, world)
    if packed.shape[0] < max_blocks:  # equal-sized contributions
        pad = torch.zeros((max_blocks - packed.shape[0], packed.shape[1]), dtype=packed.dtype, device=packed.device)
        packed = torch.cat([packed, pad], dim=0)
    if rank == 0:
        bufs = [torch.empty_like(packed) for _ in range(world)]
        dist.gather(packed, gather_list=bufs, dst=0, group=group)
        return _unpack_blocks_torch(bufs, width, height) if on_device else unpack_blocks(bufs, width, height)
    dist.gather(packed, gather_list=None, dst=0, group=group)
    return None


class ShardedRenderer:
    """One rank of an N-GPU render: packed render + the RCCL gather, both through the C ABI.

    `packed` is this rank's device buffer (1/N of a frame), `full` the assembled frame on rank 0 (else None)."""

    def __init__(self, scene, width, height, rank, world, device_index):
        self.scene, self.width, self.height, self.rank, self.world = scene, width, height, rank, world
        dev = torch.device("cuda", device_index)
        self.n_blocks = api.shard_block_count(width, height, rank, world)
        self.packed = torch.zeros((max(1, self.n_blocks), BLOCK * BLOCK, 3), dtype=torch.float32, device=dev)
        self.full = torch.zeros((height, width, 3), dtype=torch.float32, device=dev) if rank == 0 else None
        uid = None
        if world > 1:
            # the ncclUniqueId goes from rank 0 to everyone through the existing process group
            box = [None]
            if rank == 0:
                try:
                    box = [api.Comm.unique_id()]
                except Exception as e:  # noqa: BLE001 -- the other ranks are waiting in the broadcast: tell them
                    box = [e]
            dist.broadcast_object_list(box, src=0)
            if not isinstance(box[0], (bytes, bytearray)):
                raise RuntimeError("rank 0 could not draw an RCCL unique id: %s" % (box[0],))
            uid = box[0]
        self.comm = api.Comm.create(uid, rank, world, device_index)

    def params(self, spp, seed, policy="chunk", chunk=0, counters=False):
        return api.Scene.params(self.width, self.height, spp, seed, policy, chunk=chunk, counters=counters,
                                shard=(self.rank, self.world), packed=True)

    def render(self, params, stream=None, want_stats=False):
        return self.scene.render_device(self.packed.data_ptr(), params, stream=stream, want_stats=want_stats)

    def gather(self, stream=None):
        self.comm.gather(self.packed.data_ptr(), self.full.data_ptr() if self.full is not None else 0,
                         self.width, self.height, stream=stream)
        return self.full
