"""Multi-GPU plumbing for the hot path (one process per GPU, torch.distributed).

The fused conv shards naturally: images are independent (the reference splits
bs*oh rows over OpenMP threads, /root/reference/src/op_conv.cc:155-156) and the
53 KB of weights are replicated, so rank r simply owns a contiguous block of
images and no collective is on the conv data path.

A collective is needed only where op_concat joins channels that were produced on
different ranks: every rank all-gathers the per-rank NHWC shards {bs,h,w,C_r}
into a rank-major staging buffer (RCCL over xGMI; backend "nccl" is RCCL on ROCm)
and dfx_concat_submit_gathered interleaves the channels (+ReLU) in one pass.
"""
import numpy as np


def shard_range(n_total, rank, world):
    """Contiguous image range of `rank`, balanced like the reference's balance211
    (util/deepfusion_utils.h:191-208): the first (n_total % world) ranks get one more."""
    base, rem = divmod(n_total, world)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def gathered_offsets(bs, h, w, channels, itemsize):
    """Byte offset of each rank's shard inside the rank-major all-gather buffer and the
    buffer size.  Shards are padded to the largest one (16-byte rounded) so that ONE
    all_gather_into_tensor moves everything; dfx_concat_submit_gathered takes arbitrary
    16-byte-aligned offsets, so the padding is never copied out."""
    sizes = [bs * h * w * c * itemsize for c in channels]
    slot = (max(sizes) + 15) // 16 * 16
    return [r * slot for r in range(len(channels))], slot * len(channels)


def allgather_shards(local, channels, group=None):
    """All-gather per-rank NHWC tensors {bs,h,w,channels[rank]} into one flat rank-major
    byte tensor on every rank with a single collective (xGMI links are point-to-point:
    fewer, larger messages win).  Layout: gathered_offsets()."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    bs, h, w, _ = local.shape
    offs, total = gathered_offsets(bs, h, w, channels, local.element_size())
    slot = total // world
    flat = local.contiguous().view(torch.uint8).reshape(-1)
    if flat.numel() != slot:
        padded = torch.zeros(slot, dtype=torch.uint8, device=flat.device)
        padded[:flat.numel()] = flat
        flat = padded
    out = torch.empty(total, dtype=torch.uint8, device=flat.device)
    dist.all_gather_into_tensor(out, flat, group=group)
    return out
