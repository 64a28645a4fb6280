"""Multi-GPU: one process per GPU, clips sharded across ranks, outputs gathered to rank 0.

The reference has no distributed code at all: its EncodeThread drains ONE queue of videos on ONE
device (backend/workthreads.py:276-348).  The path shards naturally by clip (each video is encoded and classified
independently), so ranks never exchange data on the hot path.  The only exchange is the end-of-clip *gather* of the
output rows - (N_i, D) fp16 CLS and (N_i, C) fp32 probabilities - to the rank that writes the ``_cls.h5`` /
``_outputs.csv`` files: grouped point-to-point sends (``batch_isend_irecv`` = one ncclGroupStart/End on RCCL), so
every peer uses its own xGMI link to rank 0 and no rank receives rows it does not need.  On MI355X that is RCCL
(``backend="nccl"``); the CPU tests use ``gloo``.

``encode_files`` is the product entry point: what the reference's queue does for a list of videos, on N GPUs.
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
        # RCCL exchanges device buffers between the ranks' processes through HIP IPC handles; this pool's host driver
        # only supports the dmabuf form, and with the legacy mode RCCL fails in hipIpcGetMemHandle (invalid argument)
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
    """Variable-length gather to ``dst``.  ``local`` is this rank's list of 2-D tensors (one per local clip, all with
    the same trailing dim and dtype across ranks).  Returns on ``dst`` a list over ranks of lists of tensors (on the
    same device as the inputs); ``None`` elsewhere.

    Two steps: (1) a small all_gather of the per-clip row counts / width / dtype (control plane, a few hundred bytes);
    (2) ONE message per rank with its rows concatenated, posted as a group: ``dst`` posts world-1 receives into
    exact-size buffers, every other rank one send.  On RCCL the group is a single ncclGroupStart/End, i.e. 7
    concurrent transfers over 7 different xGMI links at 8 GPUs; nothing is padded and nothing is broadcast."""
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    if world == 1:
        return [list(local)]
    gloo = dist.get_backend() == "gloo"
    if gloo:                                    # gloo moves host memory: stage device tensors through the CPU
        local = [t.cpu() for t in local]
    ref = local[0] if len(local) else None
    device = ref.device if ref is not None else torch.device("cpu" if gloo else "cuda")
    # 1. metadata: [n_clips, width, dtype code, rows of clip 0, rows of clip 1, ...] padded to the largest clip count
    n_local = torch.tensor([len(local)], dtype=torch.int64, device=device)
    all_n = [torch.zeros_like(n_local) for _ in range(world)]
    dist.all_gather(all_n, n_local)
    max_clips = max(int(t.item()) for t in all_n)
    meta = torch.full((max_clips + 2,), -1, dtype=torch.int64, device=device)
    for i, t in enumerate(local):
        meta[i] = t.shape[0]
    if len(local):
        meta[max_clips] = local[0].shape[1]
        meta[max_clips + 1] = _DT.index(local[0].dtype)
    all_meta = [torch.zeros_like(meta) for _ in range(world)]
    dist.all_gather(all_meta, meta)
    all_meta = [m.cpu() for m in all_meta]
    width = max(int(m[max_clips].item()) for m in all_meta)
    dt_code = max(int(m[max_clips + 1].item()) for m in all_meta)
    if width < 0:                               # no rank has any clip
        return [[] for _ in range(world)] if rank == dst else None
    dtype = _DT[dt_code]
    counts = [[int(c) for c in m[:max_clips].tolist() if c >= 0] for m in all_meta]
    totals = [sum(c) for c in counts]
    # 2. one grouped point-to-point exchange
    if rank != dst:
        if totals[rank] > 0:
            payload = torch.cat(list(local), dim=0).contiguous()
            for req in dist.batch_isend_irecv([dist.P2POp(dist.isend, payload, dst)]):
                req.wait()
        return None
    bufs = {r: torch.empty((totals[r], width), dtype=dtype, device=device) for r in range(world) if r != dst and totals[r] > 0}
    if bufs:
        for req in dist.batch_isend_irecv([dist.P2POp(dist.irecv, b, r) for r, b in bufs.items()]):
            req.wait()
    out: List[List[torch.Tensor]] = []
    for r in range(world):
        if r == dst:
            out.append(list(local))
            continue
        rows, off = [], 0
        for c in counts[r]:
            rows.append(bufs[r][off:off + c] if c else torch.empty((0, width), dtype=dtype, device=device))
            off += c
        out.append(rows)
    return out


def encode_files(paths: Sequence[str], encoder, head=None, dataset_name: Optional[str] = None,
                 behaviors: Optional[Sequence[str]] = None, temperature: float = 1.0,
                 progress_callback=None) -> Optional[List[dict]]:
    """Drain a list of videos on all ranks: what the reference's EncodeThread queue (+ ClassificationThread when a model
    is live) does on one device (backend/workthreads.py:276-348, 453-519), sharded by clip.

    Clip i belongs to rank i mod world (``shard_clips``).  The ranks walk the list in rounds of ``world`` clips: every
    rank encodes its clip of the round with ``encode_file``'s chunk loop (and classifies it with ``infer_file``'s
    window loop when ``head`` is given), then the rows are gathered to rank 0 (``gather_rows``), which writes
    ``<video>_cls.h5`` (and ``<video>_<dataset_name>_outputs.csv``) in clip order with the reference's file semantics
    (``.tmp`` + rename, encoder stamp, CSV header = behaviours).  A clip that fails on its rank is logged there and
    skipped, as EncodeThread does (workthreads.py:334-336); a video without frames yields no file.

    Returns on rank 0 one record per clip {"path", "frames", "cls_file", "csv_file", "status"}, status in
    {"ok", "empty", "failed"}; ``None`` on the other ranks.  Every rank must call it with the same ``paths``."""
    import numpy as np
    from . import pipeline as P
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    gloo = dist.is_initialized() and dist.get_backend() == "gloo"
    dev = torch.device("cpu") if (gloo or not torch.cuda.is_available()) else torch.device("cuda", torch.cuda.current_device())
    paths = list(paths)
    if head is not None and (dataset_name is None or behaviors is None):
        raise ValueError("classification needs dataset_name and behaviors (infer_file's arguments)")
    D = encoder.config.hidden_size
    results: List[dict] = []
    for base in range(0, len(paths), world):
        mine = base + rank
        rows = probs = None
        status = 3                                              # 0 ok, 1 empty video, 2 failed, 3 no clip this round
        if mine < len(paths):
            try:
                rows = P.encode_rows(encoder, paths[mine], progress_callback)
                if rows is None:
                    status = 1
                else:
                    if head is not None and rows.shape[0] > 0:
                        hdev = encoder.device if hasattr(encoder, "device") else dev
                        probs = head.infer_clip(torch.from_numpy(rows).to(hdev), float(temperature)).cpu().numpy()
                    status = 0
            except Exception as e:  # noqa: BLE001 - the queue survives a bad file (workthreads.py:334-336)
                print(f"ERROR during encoding for {paths[mine]} on rank {rank}: {e}")
                rows = probs = None
                status = 2
        # round status of every rank (control plane), then the two row gathers
        st = torch.tensor([status], dtype=torch.int64, device=dev)
        if world > 1:
            all_st = [torch.zeros_like(st) for _ in range(world)]
            dist.all_gather(all_st, st)
            all_st = [int(t.item()) for t in all_st]
        else:
            all_st = [status]
        ok = status == 0
        g_rows = gather_rows([torch.from_numpy(rows).to(dev)] if ok else [], dst=0)
        g_probs = gather_rows([torch.from_numpy(probs).to(dev)] if ok and probs is not None else [], dst=0) if head is not None else None
        if rank != 0:
            continue
        for r in range(world):
            clip = base + r
            if clip >= len(paths):
                break
            rec = {"path": paths[clip], "frames": 0, "cls_file": None, "csv_file": None,
                   "status": {0: "ok", 1: "empty", 2: "failed"}[all_st[r]]}
            if all_st[r] == 0:
                try:
                    r16 = g_rows[r][0].cpu().numpy()
                    rec["frames"] = int(r16.shape[0])
                    rec["cls_file"] = P.write_cls_file(paths[clip], r16)
                    print(f"Successfully encoded {os.path.basename(paths[clip])} to {os.path.basename(rec['cls_file'])}")
                    if head is not None and g_probs[r]:
                        csv = rec["cls_file"].replace("_cls.h5", f"_{dataset_name}_outputs.csv")
                        P.write_probs_csv(csv, g_probs[r][0].cpu().numpy(), list(behaviors))
                        rec["csv_file"] = csv
                except Exception as e:  # noqa: BLE001
                    print(f"ERROR writing outputs of {paths[clip]}: {e}")
                    rec["status"] = "failed"
            results.append(rec)
    return results if rank == 0 else None


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
