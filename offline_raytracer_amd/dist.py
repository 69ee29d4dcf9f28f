"""Multi-GPU plumbing for the render call: one process per GPU, image blocks sharded
round-robin, ONE gather of the final framebuffer to rank 0 (RCCL over xGMI when the
process group's backend is "nccl"; "gloo" on CPU for the tests).

The path shards without any data-path collective: the scene (a few MB) is replicated,
every rank renders the 8x8-pixel blocks with block_id % world == rank (cost per pixel
varies ~10x across the image, so interleaving balances the ranks), and the seeding
policies are per pixel, so the union of the shards is bit-identical to a one-GPU render.
The reference has no counterpart (it is a single shared-memory process,
code/macos_main.mm:565-671).  torch is used for device memory and the collective only.
"""
import torch
import torch.distributed as dist

BLOCK = 8  # must match the kernel's implicit job space (ort_kernels.hip: 8x8 blocks)


def block_grid(width, height):
    return (width + BLOCK - 1) // BLOCK, (height + BLOCK - 1) // BLOCK


def my_block_ids(width, height, rank, world, device=None):
    bw, bh = block_grid(width, height)
    return torch.arange(rank, bw * bh, world, device=device)


def pack_blocks(image, rank, world):
    """image: [H, W, 3] float32 tensor -> [n_my_blocks, 8*8*3] (this rank's blocks, row-major block order)."""
    height, width, _ = image.shape
    bw, bh = block_grid(width, height)
    padded = image
    if bw * BLOCK != width or bh * BLOCK != height:
        padded = torch.zeros((bh * BLOCK, bw * BLOCK, 3), dtype=image.dtype, device=image.device)
        padded[:height, :width] = image
    blocks = padded.view(bh, BLOCK, bw, BLOCK, 3).permute(0, 2, 1, 3, 4).reshape(bw * bh, BLOCK * BLOCK * 3)
    return blocks.index_select(0, my_block_ids(width, height, rank, world, image.device)).contiguous()


def unpack_blocks(packed_per_rank, width, height):
    """inverse of pack_blocks over all ranks -> [H, W, 3]."""
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
    """The single collective of the path.  Every rank passes its full-size framebuffer (only
    its own blocks are meaningful); rank 0 gets the assembled [H, W, 3] image, others None."""
    if world == 1:
        return local_image
    bw, bh = block_grid(width, height)
    max_blocks = (bw * bh + world - 1) // world
    packed = pack_blocks(local_image, rank, world)
    if packed.shape[0] < max_blocks:  # equal-sized contributions
        pad = torch.zeros((max_blocks - packed.shape[0], packed.shape[1]), dtype=packed.dtype, device=packed.device)
        packed = torch.cat([packed, pad], dim=0)
    if rank == 0:
        bufs = [torch.empty_like(packed) for _ in range(world)]
        dist.gather(packed, gather_list=bufs, dst=0, group=group)
        return unpack_blocks(bufs, width, height)
    dist.gather(packed, gather_list=None, dst=0, group=group)
    return None
