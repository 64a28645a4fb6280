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


# ------------------------------------------------------------------------------------------------
# One long clip on N GPUs (last sentence): contiguous frame ranges per rank; the head's
# window of frame i needs CLS rows i-half .. i+half, so neighbouring ranks exchange `half` = seq_len // 2
# rows at each cut.  This is the one real exchange step of the path: 15 rows x 768 fp16 = 23 KB per cut.
# ------------------------------------------------------------------------------------------------
def shard_frames(n_frames: int, world: int, rank: int) -> Tuple[int, int]:
    """Balanced contiguous ranges: rank r encodes frames [start, stop)."""
    base, extra = divmod(n_frames, world)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def exchange_halo(rows: torch.Tensor, half: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """``rows`` = this rank's (n_local, D) CLS rows of its contiguous frame range (ranks in frame order).
    Returns (left, right): up to ``half`` rows preceding / following the range, taken from as many
    neighbouring ranks as needed (a rank may hold fewer than ``half`` frames); empty at the clip's ends.

    One all_gather of a fixed (2*half + 1, D)-shaped block per rank: its first and last min(n_local, half)
    rows and n_local (the blocks are tiny, so the all_gather is cheaper than a send/recv schedule)."""
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    D = rows.shape[1]
    if world == 1 or half == 0:
        return rows[:0], rows[:0]
    gloo = dist.get_backend() == "gloo"
    src = rows.cpu() if gloo else rows
    n = src.shape[0]
    k = min(n, half)
    blk = torch.zeros((2 * half + 1, D), dtype=torch.float32, device=src.device)
    if k:
        blk[:k] = src[:k].float()                    # head rows (fp16 -> fp32 is exact)
        blk[half:half + k] = src[n - k:].float()     # tail rows
    blk[2 * half, 0] = float(n)                      # row count (exact in fp32 up to 2^24 frames per rank)
    allb = [torch.empty_like(blk) for _ in range(world)]
    dist.all_gather(allb, blk)
    counts = [int(b[2 * half, 0].item()) for b in allb]

    def collect(order, take_tail: bool):
        parts, need = [], half
        for r in order:
            if need <= 0:
                break
            kr = min(counts[r], half)
            if kr == 0:
                continue
            t = allb[r][half:half + kr] if take_tail else allb[r][:kr]
            take = min(need, kr)
            parts.append(t[kr - take:] if take_tail else t[:take])
            need -= take
        return parts

    left_parts = collect(range(rank - 1, -1, -1), True)          # nearest neighbour first
    right_parts = collect(range(rank + 1, world), False)
    left = torch.cat(left_parts[::-1]) if left_parts else blk[:0]
    right = torch.cat(right_parts) if right_parts else blk[:0]
    return left.to(rows.dtype).to(rows.device), right.to(rows.dtype).to(rows.device)


def classify_sharded(head, local_cls16: torch.Tensor, temperature: float = 1.0) -> torch.Tensor:
    """Probabilities (n_local, C) for this rank's frame range of ONE clip split by ``shard_frames``:
    halo exchange, then the range form of the head (windows clamp only at the true clip ends)."""
    half = head.seq_len // 2
    left, right = exchange_halo(local_cls16, half)
    buf = torch.cat([left, local_cls16, right]).contiguous()
    n_local = local_cls16.shape[0]
    probs = torch.empty((buf.shape[0], head.out_features), dtype=torch.float32, device=buf.device)
    if n_local:
        head.infer_range_into(buf, buf.shape[0], left.shape[0], n_local, probs, temperature)
    return probs[left.shape[0]:left.shape[0] + n_local]
