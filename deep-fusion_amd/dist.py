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
    """Byte offset of each rank's shard inside the rank-major all-gather buffer
    (variable channel counts allowed) and the total size."""
    sizes = [bs * h * w * c * itemsize for c in channels]
    offs = np.concatenate([[0], np.cumsum(sizes)[:-1]]).astype(np.uint64)
    return [int(o) for o in offs], int(sum(sizes))


def allgather_shards(local, channels, group=None):
    """All-gather per-rank NHWC tensors {bs,h,w,channels[rank]} into one flat
    rank-major byte tensor on every rank.  Equal channel counts use a single
    all_gather_into_tensor (one large collective: xGMI links are point-to-point,
    so fewer, larger messages win); ragged counts fall back to all_gather."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    flat = local.contiguous().view(torch.uint8).reshape(-1)
    if len(set(channels)) == 1:
        out = torch.empty(flat.numel() * world, dtype=torch.uint8, device=flat.device)
        dist.all_gather_into_tensor(out, flat, group=group)
        return out
    bs, h, w, _ = local.shape
    per_px = local.element_size()
    parts = [torch.empty(bs * h * w * c * per_px, dtype=torch.uint8, device=flat.device)
             for c in channels]
    dist.all_gather(parts, flat, group=group)
    return torch.cat(parts)
