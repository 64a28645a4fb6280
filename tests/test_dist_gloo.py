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


def _start_without_gpu(procs):
    """Start the CPU ranks with no GPU visible to them: these are CPU tests (stand-in encoders, gloo), and on a GPU box every
    child that imports torch would otherwise open the device - eight of them trip the box's limit of processes on one GPU."""
    keys = ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES")
    saved = {k: os.environ.get(k) for k in keys}
    try:
        os.environ["HIP_VISIBLE_DEVICES"] = os.environ["CUDA_VISIBLE_DEVICES"] = ""
        for p in procs:
            p.start()
    finally:
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v



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
    _start_without_gpu(procs)
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


# ---- one clip split across ranks: halo exchange ---------------------------------------------------
def _halo_worker(rank, world, port, q, n_frames, half):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    cdist.init_from_env("gloo")
    full = (torch.arange(n_frames)[:, None] * 4 + torch.arange(4)[None, :]).to(torch.float16)     # row i = [4i .. 4i+3]
    a, b = cdist.shard_frames(n_frames, world, rank)
    left, right = cdist.exchange_halo(full[a:b].clone(), half)
    want_l = full[max(0, a - half):a]
    want_r = full[b:min(n_frames, b + half)]
    ok = left.dtype == torch.float16 and torch.equal(left, want_l) and torch.equal(right, want_r)
    q.put((rank, bool(ok), (a, b)))
    dist.destroy_process_group()


@pytest.mark.parametrize("world,n_frames,half", [(2, 100, 15), (3, 20, 15), (3, 7, 15), (2, 1, 15)])
def test_halo_exchange_gloo(world, n_frames, half):
    """Shards shorter than the halo (20 frames on 3 ranks, 7 on 3, 1 on 2) pull rows from several neighbours."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_halo_worker, args=(r, world, port, q, n_frames, half)) for r in range(world)]
    _start_without_gpu(procs)
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    res = sorted(q.get(timeout=5) for _ in range(world))
    assert all(ok for _, ok, _ in res), res
    assert res[0][2][0] == 0 and res[-1][2][1] == n_frames
    assert all(res[i][2][1] == res[i + 1][2][0] for i in range(world - 1))


def test_shard_frames_partition():
    for n, w in ((18000, 8), (10, 3), (2, 4), (0, 2)):
        cuts = [cdist.shard_frames(n, w, r) for r in range(w)]
        assert cuts[0][0] == 0 and cuts[-1][1] == n and all(cuts[i][1] == cuts[i + 1][0] for i in range(w - 1))
        sizes = [b - a for a, b in cuts]
        assert max(sizes) - min(sizes) <= 1


@pytest.mark.gpu
def test_sharded_clip_classification_equals_whole_clip():
    """Frame-range shards + halos through the range form of the head give the whole-clip probabilities bit
    for bit (the exchange itself is covered on gloo above; here the halos are cut from the full clip)."""
    from cbas_amd import config as C, synth, weights as W
    from cbas_amd.head import ClassifierLSTMDeltas
    head = ClassifierLSTMDeltas(768, 9)
    head.load_state_dict(W.synth_head_weights(C.HeadConfig(), 4321))
    head.to("cuda")
    cls = torch.from_numpy(synth.cls_walk(9, 500, 768)).cuda()
    whole = head.infer_clip(cls)
    half = head.seq_len // 2
    for world in (1, 3, 8):
        out = []
        for r in range(world):
            a, b = cdist.shard_frames(500, world, r)
            left, right = cls[max(0, a - half):a], cls[b:min(500, b + half)]
            buf = torch.cat([left, cls[a:b], right]).contiguous()
            probs = torch.empty((buf.shape[0], 9), dtype=torch.float32, device="cuda")
            head.infer_range_into(buf, buf.shape[0], left.shape[0], b - a, probs)
            out.append(probs[left.shape[0]:left.shape[0] + b - a])
        torch.cuda.synchronize()
        assert torch.equal(torch.cat(out), whole), world
    # world == 1 through the public helper
    assert torch.equal(cdist.classify_sharded(head, cls), whole)
    head.close()
