"""CPU-only, world_size 2 over gloo: the multi-GPU plumbing of the hot path
(deep-fusion_amd/dist.py).  The conv path shards by image with no collective; the only
exchange is the all-gather in front of op_concat.  The gathered
rank-major buffer is checked against the oracle's concat(+relu)."""
import importlib
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ddist = importlib.import_module("deep-fusion_amd.dist")


def test_shard_range_is_balanced_and_covers():
    for n, w in ((1024, 8), (128, 8), (10, 4), (3, 8), (7, 2)):
        spans = [ddist.shard_range(n, r, w) for r in range(w)]
        assert spans[0][0] == 0 and spans[-1][1] == n
        assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
        sizes = [b - a for a, b in spans]
        assert max(sizes) - min(sizes) <= 1


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, channels, out):
    sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import oracle as orc
    bs, h, w = 3, 5, 4
    rng = np.random.default_rng(100)                      # same stream on every rank
    shards = [rng.integers(-128, 128, (bs, h, w, c)).astype(np.int8) for c in channels]
    local = torch.from_numpy(shards[rank])
    g = ddist.allgather_shards(local, channels)
    offs, total = ddist.gathered_offsets(bs, h, w, channels, 1)
    ok = g.numel() == total
    # rank-major layout: shard r sits at offs[r]
    for r, s in enumerate(shards):
        seg = g[offs[r]:offs[r] + s.size].numpy().view(np.int8).reshape(s.shape)
        ok = ok and np.array_equal(seg, s)
    # what dfx_concat_submit_gathered must produce from that buffer
    ref = orc.concat(shards, True)
    parts = [g[offs[r]:offs[r] + shards[r].size].numpy().view(np.int8).reshape(shards[r].shape) for r in range(world)]
    ok = ok and np.array_equal(np.maximum(np.concatenate(parts, axis=3), 0), ref)
    # image sharding of the conv path: contiguous, disjoint, complete
    a, b = ddist.shard_range(11, rank, world)
    t = torch.zeros(11, dtype=torch.int32)
    t[a:b] = 1
    dist.all_reduce(t)
    ok = ok and bool((t == 1).all())
    out[rank] = bool(ok)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("channels", [[32, 32], [16, 48]])
def test_allgather_concat_world2(channels):
    port = _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(2, port, channels, out), nprocs=2, join=True)
    assert dict(out) == {0: True, 1: True}
