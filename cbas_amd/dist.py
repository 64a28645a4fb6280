"""Multi-GPU: one process per GPU, clips sharded across ranks, outputs gathered to rank 0.

The reference has no distributed code at all; the path shards naturally by clip
(each video is encoded and classified independently), so ranks never exchange data on the hot
path.  The only collective is the end-of-clip *gather* of the output rows — (N_i, D) fp16 CLS and
(N_i, C) fp32 probabilities — to the rank that writes the ``_cls.h5`` / ``_outputs.csv`` files.
On MI355X that is RCCL over xGMI (``backend="nccl"``); the CPU tests use ``gloo``.
"""
from __future__ import annotations

import os
from typing import List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist


def init_from_env(backend: Optional[str] = None) -> Tuple[int, int, int]:
    """Initialise torch.distributed from torchrun's environment; returns (rank, world, local_rank).
    A no-op single-process world when WORLD_SIZE is unset or 1."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def shard_clips(n_clips: int, world: int, rank: int) -> List[int]:
    """Round-robin clip assignment: clip i -> rank i mod world."""
    return list(range(rank, n_clips, world))


def owner_of(clip: int, world: int) -> int:
    return clip % world


def gather_rows(local: Sequence[torch.Tensor], dst: int = 0) -> Optional[List[List[torch.Tensor]]]:
    """Variable-length gather.  ``local`` is this rank's list of 2-D tensors (one per local clip,
    all with the same trailing dim and dtype across ranks).  Returns on ``dst`` a list over ranks
    of lists of tensors (on the same device as the inputs); ``None`` elsewhere.

    Collectives: all_gathers of the clip counts / row counts / dtype code, then one padded all_gather of
    the concatenated rows (payloads are a few tens of MB per 30-minute clip; every
    GPU has its own xGMI link to every peer, so the exchange is link-parallel)."""
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    if world == 1:
        return [list(local)]
    if dist.get_backend() == "gloo":          # gloo moves host memory: stage device tensors through the CPU
        local = [t.cpu() for t in local]
    ref = local[0] if len(local) else None
    device = ref.device if ref is not None else torch.device("cuda" if dist.get_backend() == "nccl" else "cpu")
    # 1. counts: (max_clips_per_rank,) per rank, -1 padded
    n_local = torch.tensor([len(local)], dtype=torch.int64, device=device)
    all_n = [torch.zeros_like(n_local) for _ in range(world)]
    dist.all_gather(all_n, n_local)
    max_clips = max(int(t.item()) for t in all_n)
    meta = torch.full((max_clips + 1,), -1, dtype=torch.int64, device=device)
    for i, t in enumerate(local):
        meta[i] = t.shape[0]
    meta[max_clips] = local[0].shape[1] if len(local) else -1
    all_meta = [torch.zeros_like(meta) for _ in range(world)]
    dist.all_gather(all_meta, meta)
    width = max(int(m[max_clips].item()) for m in all_meta)
    totals = [int(m[:max_clips].clamp(min=0).sum().item()) for m in all_meta]
    max_rows = max(totals)
    dtype = None
    for t in local:
        dtype = t.dtype
    dt_code = torch.tensor([_DT.index(dtype) if dtype is not None else -1], dtype=torch.int64, device=device)
    all_dt = [torch.zeros_like(dt_code) for _ in range(world)]
    dist.all_gather(all_dt, dt_code)
    dtype = _DT[max(int(t.item()) for t in all_dt)]
    # 2. padded gather of the concatenated rows
    buf = torch.zeros((max_rows, width), dtype=dtype, device=device)
    if len(local):
        cat = torch.cat(list(local), dim=0)
        buf[:cat.shape[0]] = cat
    # all_gather rather than gather: the most widely supported collective on every backend (RCCL
    # gather is emulated with send/recv anyway); the payload is a few tens of MB per clip
    recv = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(recv, buf)
    if rank != dst:
        return None
    out: List[List[torch.Tensor]] = []
    for r in range(world):
        counts = [int(c) for c in all_meta[r][:max_clips].tolist() if c >= 0]
        rows, off = [], 0
        for c in counts:
            rows.append(recv[r][off:off + c])
            off += c
        out.append(rows)
    return out


_DT = [torch.float16, torch.float32, torch.uint8, torch.int64, torch.bfloat16]


def interleave_by_clip(per_rank: List[List[torch.Tensor]], n_clips: int) -> List[torch.Tensor]:
    """Undo ``shard_clips``: per_rank[r][j] is clip r + j*world -> list indexed by clip id."""
    world = len(per_rank)
    out: List[Optional[torch.Tensor]] = [None] * n_clips
    for r in range(world):
        for j, t in enumerate(per_rank[r]):
            out[r + j * world] = t
    assert all(o is not None for o in out)
    return out  # type: ignore[return-value]


def barrier() -> None:
    if dist.is_initialized():
        if dist.get_backend() == "nccl":
            dist.barrier(device_ids=[torch.cuda.current_device()])
        else:
            dist.barrier()


def max_over_ranks(value: float, device) -> float:
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return value
    if dist.get_backend() == "gloo":
        device = torch.device("cpu")
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
