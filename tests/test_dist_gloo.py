"""World-size-2 rehearsal of the multi-GPU path on CPU (gloo): clip sharding, the variable-length
gather of output rows to rank 0, timing reduction."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from cbas_amd import dist as cdist


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    r, w, _ = cdist.init_from_env("gloo")
    assert (r, w) == (rank, world)
    n_clips = 5
    lengths = [7, 0 + 3, 11, 1, 6]                      # ragged clips
    mine = cdist.shard_clips(n_clips, world, rank)
    cls = [torch.full((lengths[c], 8), float(c), dtype=torch.float16) + torch.arange(lengths[c])[:, None].half() / 64
           for c in mine]
    probs = [torch.full((lengths[c], 3), 0.25 * c, dtype=torch.float32) for c in mine]
    g_cls = cdist.gather_rows(cls, dst=0)
    g_pr = cdist.gather_rows(probs, dst=0)
    t = cdist.max_over_ranks(1.0 + rank, torch.device("cpu"))
    cdist.barrier()
    if rank == 0:
        by_clip = cdist.interleave_by_clip(g_cls, n_clips)
        pr_clip = cdist.interleave_by_clip(g_pr, n_clips)
        ok = all(by_clip[c].shape == (lengths[c], 8) and pr_clip[c].shape == (lengths[c], 3) for c in range(n_clips))
        ok &= all(float(by_clip[c][0, 0]) == float(c) for c in range(n_clips))
        ok &= all(torch.allclose(pr_clip[c], torch.full((lengths[c], 3), 0.25 * c)) for c in range(n_clips))
        ok &= by_clip[2].dtype == torch.float16 and t == float(world)
        q.put(bool(ok))
    else:
        assert g_cls is None and g_pr is None
    dist.destroy_process_group()


def test_two_rank_gather_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert q.get(timeout=5) is True


def test_shard_clips_partition():
    for n, w in ((8, 8), (5, 2), (3, 4), (0, 2)):
        seen = sorted(c for r in range(w) for c in cdist.shard_clips(n, w, r))
        assert seen == list(range(n))
        assert all(cdist.owner_of(c, w) == r for r in range(w) for c in cdist.shard_clips(n, w, r))


def test_single_process_gather_is_identity():
    t = [torch.zeros(3, 4)]
    assert cdist.gather_rows(t)[0][0] is t[0]
