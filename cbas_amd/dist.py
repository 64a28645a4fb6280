"""Multi-GPU: one process per GPU, clips sharded across ranks, outputs gathered to rank 0.

The reference has no distributed code at all: its EncodeThread drains ONE queue of videos on ONE
device (backend/workthreads.py:276-348).  The path shards naturally by clip (each video is encoded and classified
independently), so ranks never exchange data on the hot path.  The only exchange is the end-of-clip *gather* of the
output rows - (N_i, D) fp16 CLS and (N_i, C) fp32 probabilities - to the rank that writes the ``_cls.h5`` /
``_outputs.csv`` files: grouped point-to-point sends (``batch_isend_irecv`` = one ncclGroupStart/End on RCCL), so
every peer uses its own xGMI link to rank 0 and no rank receives rows it does not need.  On MI355X that is RCCL
(``backend="nccl"``); the CPU tests use ``gloo``.

``encode_files`` is the product entry point: what the reference's queue does for a list of videos, on N GPUs.

Status of the transports: the gloo path runs in the CPU test-suite (world 2, 3, 4) and with two gloo ranks sharing one
GPU on the real kernels; the RCCL path has not run on a multi-GPU node yet (none was available to the builder) - it is
the same code with device tensors, kept to grouped point-to-point calls on the world communicator.
"""
from __future__ import annotations

import os
from typing import List, Optional, Sequence, Tuple

import numpy as np
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
            # more ranks than GPUs cannot work on RCCL (one communicator rank per device): say so here, not in a hang
            ndev = torch.cuda.device_count()
            if int(os.environ.get("LOCAL_WORLD_SIZE", world)) > max(1, ndev):
                raise RuntimeError(f"{world} ranks for {ndev} GPU(s): the RCCL backend needs one GPU per rank "
                                   "(use CBAS_DIST_BACKEND=gloo to rehearse with ranks sharing a GPU)")
            torch.cuda.set_device(local % max(1, ndev))
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


class _ClipQueue:
    """The shared work queue: ``next()`` hands out each clip index exactly once across all ranks (the reference's
    ``gui_state.encode_tasks`` list popped under ``encode_lock``, backend/workthreads.py:284-290, made node-wide).  It is a
    counter in the process group's key-value store, so a rank that is given a long clip simply takes fewer of them; no
    collective is involved and no rank waits for another."""

    def __init__(self, n: int, store, prefix: str):
        self.n, self._store, self._key, self._local = int(n), store, prefix + "next", 0

    def next(self) -> Optional[int]:
        if self._store is None:
            i, self._local = self._local, self._local + 1
        else:
            i = int(self._store.add(self._key, 1)) - 1
        return i if i < self.n else None

    def remaining(self) -> int:
        """Clips nobody has taken yet (a snapshot: other ranks keep taking them)."""
        taken = self._local if self._store is None else int(self._store.add(self._key, 0))
        return max(0, self.n - taken)


class _OutputWriter:
    """Rank 0's file writers, off the encode loop's critical path: ``_cls.h5`` files on ONE thread (libhdf5 is not
    thread-safe), ``_outputs.csv`` files on a small pool (the native formatter releases the GIL).  ``submit`` only
    queues; records are completed by the threads; ``close`` waits for everything."""

    def __init__(self, records: List[dict], dataset_name, behaviors, attrs: dict, csv_threads: int = 2, max_pending: int = 16):
        import queue
        import threading
        from concurrent.futures import ThreadPoolExecutor
        self._records, self._name, self._behaviors, self._attrs = records, dataset_name, behaviors, attrs
        self._q: "queue.Queue" = queue.Queue(maxsize=max_pending)
        self._csv = ThreadPoolExecutor(max_workers=max(1, csv_threads), thread_name_prefix="cbas-csv")
        self._futs: list = []
        self._t = threading.Thread(target=self._h5_loop, name="cbas-h5-writer", daemon=True)
        self._t.start()

    def submit(self, clip: int, rows, probs) -> None:
        from . import pipeline as P
        rec = self._records[clip]
        rec["frames"] = int(rows.shape[0])
        if probs is not None and self._name is not None:
            csv = os.path.splitext(rec["path"])[0] + f"_{self._name}_outputs.csv"
            self._futs.append(self._csv.submit(self._write_csv, rec, csv, probs, P))
        self._q.put((rec, rows))                    # blocks when max_pending clips are waiting: back-pressure

    @staticmethod
    def _fail(rec, what, e):
        print(f"ERROR writing {what} of {rec['path']}: {e}")
        rec["status"] = "failed"

    def _write_csv(self, rec, csv, probs, P):
        try:
            P.write_probs_csv(csv, probs, list(self._behaviors))
            rec["csv_file"] = csv
        except Exception as e:  # noqa: BLE001
            self._fail(rec, "the probabilities", e)

    def _h5_loop(self):
        from . import pipeline as P
        while True:
            item = self._q.get()
            if item is None:
                return
            rec, rows = item
            try:
                rec["cls_file"] = P.write_cls_file(rec["path"], rows, self._attrs)
                print(f"Successfully encoded {os.path.basename(rec['path'])} to {os.path.basename(rec['cls_file'])}")
            except Exception as e:  # noqa: BLE001
                self._fail(rec, "the CLS rows", e)

    def close(self) -> None:
        self._q.put(None)
        self._t.join()
        for f in self._futs:
            f.result()
        self._csv.shutdown()
        for rec in self._records:                   # a clip whose CSV or rows failed to be written has no output at all
            if rec["status"] == "failed":
                rec["cls_file"] = rec["csv_file"] = None


_ef_calls = 0
_ST_OK, _ST_EMPTY, _ST_FAILED = 0, 1, 2


class _PeerLost(RuntimeError):
    """The other side of a transfer will not complete it (its rank has gone silent, or rank 0 gave the gather up)."""


def _wait_done(work, give_up=None) -> None:
    """Host-side completion of a transfer.  ``Work.wait()`` on RCCL only makes the current torch stream wait; the buffers
    these transfers read are rewritten by kernels on the library's own streams, so the host has to see them finished.
    ``give_up()`` is polled about four times a second while the transfer is outstanding: when it returns a reason the wait
    is abandoned with ``_PeerLost`` (the transfer itself cannot be cancelled; the caller stops using the communicator)."""
    import time
    if give_up is None:
        work.wait()
        while not work.is_completed():
            time.sleep(0.0002)
        return
    if dist.get_backend() != "nccl":
        # gloo: is_completed() only turns true inside wait(), and wait() blocks the host - so the wait runs on a helper thread
        # (left behind, blocked, if the transfer is given up: the job is ending then)
        import threading
        done = threading.Event()
        err: list = []

        def waiter():
            try:
                work.wait()
            except BaseException as e:  # noqa: BLE001
                err.append(e)
            done.set()
        threading.Thread(target=waiter, name="cbas-p2p-wait", daemon=True).start()
        while not done.wait(0.25):
            why = give_up()
            if why:
                raise _PeerLost(why)
        if err:
            raise err[0]
        return
    n = 0
    while not work.is_completed():         # RCCL: wait() would only make the current stream wait; poll, about 4 give_up()s a second
        time.sleep(0.0002)
        n += 1
        if n % 1250 == 0:
            why = give_up()
            if why:
                raise _PeerLost(why)
    work.wait()                            # completed: orders the current stream after it, returns at once


class _SettleOnce:
    """A transfer's host-side completion, waited for at most once however many owners ask (the session whose buffers it reads
    asks before it is reused and when it is closed; the rank asks before it leaves).  A second ``wait()`` on a finished gloo
    send never returns - found by the device-rows rehearsal of the RCCL ranks' control flow (tests/test_gpu_round2.py)."""

    def __init__(self, work, give_up):
        self.work, self.give_up, self.done = work, give_up, False

    def __call__(self):
        if not self.done:
            _wait_done(self.work, self.give_up)
            self.done = True


class _DictStore:
    """The four store calls encode_files uses, on a dict: the world of one process has no process group store."""

    def __init__(self):
        import threading
        self._d, self._lock = {}, threading.Lock()

    def set(self, k, v):
        with self._lock:
            self._d[k] = v if isinstance(v, bytes) else str(v).encode()

    def get(self, k):
        with self._lock:
            return self._d[k]

    def add(self, k, n):
        with self._lock:
            v = int(self._d.get(k, b"0")) + int(n)
            self._d[k] = str(v).encode()
            return v

    def check(self, keys):
        with self._lock:
            return all(k in self._d for k in keys)


class _Liveness:
    """Rank 0's view of which ranks can still publish or complete a transfer.  A rank has LEFT (it sets ``left<r>`` after its
    last ticket) or has gone SILENT: the value of its heartbeat key has not CHANGED for ``dead_after`` seconds of this
    process's own monotonic clock - no wall-clock time of another host is compared with ours."""

    def __init__(self, st, prefix: str, world: int, dead_after: float):
        import time
        self._st, self._prefix, self._dead_after = st, prefix, dead_after
        now = time.monotonic()
        self._last = {r: (None, now) for r in range(world)}
        self._left: set = set()

    def left(self, r: int) -> bool:
        if r not in self._left and self._st.check([f"{self._prefix}left{r}"]):
            self._left.add(r)
        return r in self._left

    def silent(self, r: int) -> bool:
        import time
        key = f"{self._prefix}hb{r}"
        now = time.monotonic()
        val = self._st.get(key) if self._st.check([key]) else None
        seen, since = self._last[r]
        if val != seen:
            self._last[r] = (val, now)
            return False
        return now - since > self._dead_after


def _p2p(op, tensor: torch.Tensor, peer: int):
    """One point-to-point transfer, posted as a group of one (on RCCL: ncclGroupStart / ncclSend|ncclRecv / ncclGroupEnd on
    the world communicator - no per-pair communicator is created); returns the work handles."""
    return dist.batch_isend_irecv([dist.P2POp(op, tensor, peer)])


def encode_files(paths: Sequence[str], encoder, head=None, dataset_name: Optional[str] = None,
                 behaviors: Optional[Sequence[str]] = None, temperature: float = 1.0,
                 progress_callback=None, csv_threads: int = 2, local_writes: bool = False) -> Optional[List[dict]]:
    """Drain a list of videos on all ranks: what the reference's EncodeThread queue (+ ClassificationThread when a model
    is live) does on one device (backend/workthreads.py:276-348, 453-519), sharded by clip over one process per GPU.

    * **Queue.**  Every rank pulls the next clip index from a shared counter (``_ClipQueue``) when it is free - not a
      static i mod N split, so one long clip does not stall the other GPUs.
    * **Clip.**  ``pipeline.ClipRunner``: the chunk loop of ``encode_file`` feeding the window loop of ``infer_file``
      inside the fused native session; the rows stay in HBM.
    * **Gather.**  When a clip is done its rank publishes a ticket {rank, clip, rows, status} in the store and posts ONE
      send of the fp16 rows (+ one of the probabilities) to rank 0 - straight from the session's device buffers on RCCL -
      and goes on with its next clip (two sessions alternate, so the send drains under the next clip's kernels).  A
      receiver thread on rank 0 takes the tickets in order and posts the matching receives: N-1 peers, N-1 xGMI links,
      nothing padded, nothing broadcast, no rank waits at a round boundary.
    * **Files.**  The receiver hands the rows to ``_OutputWriter`` (one HDF5 thread + a CSV pool); rank 0's own encode loop
      never writes.  Files are what ``encode_file`` / ``infer_file`` write, byte for byte (``.tmp`` + rename, encoder
      stamp, CSV header = behaviours), whatever rank encoded them and in whatever order they completed.

    ``local_writes=True`` (one node, one filesystem): every rank writes the files of ITS clips with its own writer threads
    and only a status ticket goes to rank 0 - no rows cross xGMI and no rank's files queue behind another's.  The gather is
    the default because it is what the path's specification asks for; it costs a tail when there is about one clip per GPU:
    all clips end together and their `_cls.h5` writes (22 ms per 18 000-frame clip) queue on rank 0's one HDF5 thread -
    8 x 22 ms after a 0.81 s encode at BASELINE configs[2] (tests/test_dist_encode_files.py, the world-8 rehearsal).

    A clip that fails on its rank is logged there and skipped, as EncodeThread does (workthreads.py:334-336); a video
    without frames yields no file.  Returns on rank 0 one record per clip, in clip order, {"path", "frames", "cls_file",
    "csv_file", "status", "rank"} with status in {"ok", "empty", "failed"}; ``None`` on the other ranks.  Every rank must call
    it with the same ``paths``."""
    import json
    import threading
    import time
    from . import pipeline as P
    global _ef_calls
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    nccl = world > 1 and dist.get_backend() == "nccl"
    # The control flow of the RCCL ranks - results left in the session's DEVICE buffers (two sessions alternating), a rank's next
    # clip started while the previous one's sends drain, no pipelining of host copies on ranks > 0 - can be rehearsed without
    # RCCL: CBAS_DIST_DEVICE_ROWS=1 on another backend takes the same branches and sends a host copy made at send time
    # (tests/test_gpu_round2.py; no multi-GPU node has run the RCCL form yet).
    devrows = nccl or (world > 1 and os.environ.get("CBAS_DIST_DEVICE_ROWS") == "1")
    paths = list(paths)
    if head is not None and (dataset_name is None or behaviors is None):
        raise ValueError("classification needs dataset_name and behaviors (infer_file's arguments)")
    D = encoder.config.hidden_size
    Cn = head.out_features if head is not None else 0
    _ef_calls += 1
    prefix = f"cbas/encode_files/{_ef_calls}/"
    store = dist.distributed_c10d._get_default_store() if world > 1 else None
    if world > 1:
        barrier()          # every rank is here; on RCCL this also creates the communicator the transfers below share
    queue_ = _ClipQueue(len(paths), store, prefix)
    tstore = store if store is not None else _DictStore()  # tickets, counters, abort flag (a dict in a world of one process)
    abort_key = prefix + "abort"
    dead_after = float(os.environ.get("CBAS_GATHER_DEAD_AFTER", "120"))
    hb_stop = threading.Event()
    hb = None
    if store is not None:
        hb_store = store.clone() if hasattr(store, "clone") else store

        def heartbeat():
            beat = 0
            while True:
                beat += 1
                try:
                    hb_store.set(f"{prefix}hb{rank}", str(beat))       # any CHANGING value: rank 0 times the changes on its own clock
                except Exception:  # noqa: BLE001 - the store is going away with the job
                    return
                if hb_stop.wait(1.0):
                    return
        hb = threading.Thread(target=heartbeat, name="cbas-heartbeat", daemon=True)
        hb.start()
    runner = P.ClipRunner(encoder, head, temperature, sessions=2 if devrows else 1)
    dev = encoder.device if nccl else torch.device("cpu")

    records = [{"path": p, "frames": 0, "cls_file": None, "csv_file": None, "status": "failed", "rank": None} for p in paths]
    writer = receiver = None
    local_done: dict = {}                                  # rank 0's own clips: clip -> (rows, probs) host arrays
    recv_err: list = []
    lw = bool(local_writes) and world > 1                  # every rank writes its own clips' files
    if rank == 0 or lw:
        writer = _OutputWriter(records, dataset_name if head is not None else None, behaviors, P.file_attrs(encoder),
                               csv_threads)
    if rank == 0:

        def take(t: dict):
            clip, src, n = t["clip"], t["rank"], t["n"]
            rec = records[clip]
            rec["rank"] = src
            rec["status"] = {_ST_OK: "ok", _ST_EMPTY: "empty", _ST_FAILED: "failed"}[t["status"]]
            if t["status"] != _ST_OK:
                return
            if t.get("local") and src != 0:                # written by its own rank (local_writes): the outcome follows
                rec["frames"] = int(n)
                return
            if src == 0:
                rows, probs = local_done.pop(clip)
            else:
                rows = torch.empty((n, D), dtype=torch.float16, device=dev)
                probs = torch.empty((n, Cn), dtype=torch.float32, device=dev) if (Cn and t["probs"]) else None
                if n:
                    # a sender that dies after publishing its ticket never completes this transfer: give up when it has left
                    # without it or gone silent (the receive cannot be cancelled: the caller abandons the gather)
                    def lost(src=src):
                        return (f"rank {src} went away with the rows of {paths[clip]} announced but not sent"
                                if live is not None and (live.silent(src)) else None)
                    for w in _p2p(dist.irecv, rows, src):
                        _wait_done(w, lost)
                    if probs is not None:
                        for w in _p2p(dist.irecv, probs, src):
                            _wait_done(w, lost)
                rows = rows.cpu().numpy()
                probs = probs.cpu().numpy() if probs is not None else None
            writer.submit(clip, rows, probs)

        def receive():
            """Tickets are per-rank sequences: rank r's j-th finished clip is `t<r>_<j>`, announced by incrementing `cnt<r>`
            AFTER the ticket is in the store - there is no shared sequence number a dying rank could leave a hole in, and
            the order of a rank's tickets is the order of its sends (what point-to-point matching needs)."""
            try:
                if nccl:
                    torch.cuda.set_device(dev)
                # its own client connection: a blocking store call here must not hold up the encode loop's counter
                st = store.clone() if store is not None and hasattr(store, "clone") else tstore
                nonlocal live
                live = _Liveness(st, prefix, world, dead_after) if store is not None else None
                nxt = [0] * world
                done_ranks: set = set()
                got, idle, gather_over = 0, 0, False
                while got < len(paths) and len(done_ranks) < world:       # exactly one ticket per clip
                    progressed = False
                    for r in range(world):
                        if r in done_ranks:
                            continue
                        # `left<r>` is set after rank r's last ticket: seen BEFORE the counter is read, the counter is final
                        is_left = live is not None and idle % 50 == 0 and live.left(r)
                        avail = int(st.add(f"{prefix}cnt{r}", 0))
                        while nxt[r] < avail:
                            t = json.loads(st.get(f"{prefix}t{r}_{nxt[r]}").decode())
                            nxt[r] += 1
                            got += 1
                            progressed = True
                            if gather_over and t["rank"] != 0 and t["status"] == _ST_OK and not t.get("local"):
                                records[t["clip"]].update(rank=t["rank"], status="failed")      # rows that will not be received
                                continue
                            try:
                                take(t)
                            except _PeerLost as e:
                                # the announced rows will never arrive and the receive cannot be cancelled (on RCCL every later
                                # receive would queue behind it): this clip and every clip of another rank not received yet are
                                # "failed", senders still waiting are told to stop; rank 0's own clips are still written
                                print(f"cbas_amd.encode_files: {e}; no further rows are gathered")
                                records[t["clip"]]["status"] = "failed"
                                tstore.set(abort_key, "1")
                                gather_over = True
                        if is_left:
                            done_ranks.add(r)
                    if progressed:
                        idle = 0
                        continue
                    if stop.is_set():
                        return
                    time.sleep(0.001)
                    idle += 1
                    # Liveness (about once a second): a rank whose heartbeat has not changed for `dead_after` seconds will
                    # publish nothing more; its outstanding clips stay "failed" in the records and the receiver goes on with
                    # the others instead of blocking rank 0 for ever.
                    if live is not None and idle % 1000 == 0:
                        for r in range(world):
                            if r not in done_ranks and not live.left(r) and live.silent(r):
                                print(f"cbas_amd.encode_files: rank {r} has not been heard from for {dead_after:.0f} s; "
                                      "its outstanding clips are marked failed")
                                done_ranks.add(r)
            except BaseException as e:  # noqa: BLE001 - re-raised by the caller after the join
                recv_err.append(e)
                # the gather is over: senders still waiting for a receive that will never be posted must stop waiting
                try:
                    tstore.set(abort_key, "1")
                except Exception:  # noqa: BLE001
                    pass

        live = None
        stop = threading.Event()
        receiver = threading.Thread(target=receive, name="cbas-gather", daemon=True)
        receiver.start()

    taken: set = set()                                     # clips this rank drew from the queue ...
    published: set = set()                                 # ... and the ones it has published a ticket for

    def publish(clip: int, status: int, n: int, has_probs: bool):
        published.add(clip)
        t = {"rank": rank, "clip": clip, "status": status, "n": int(n), "probs": bool(has_probs), "local": lw}
        tstore.set(f"{prefix}t{rank}_{clip_seq[0]}", json.dumps(t))       # the ticket first ...
        clip_seq[0] += 1
        tstore.add(f"{prefix}cnt{rank}", 1)                                # ... then its announcement: no holes

    def gather_given_up():
        return "rank 0 gave the gather up" if tstore.check([abort_key]) else None

    clip_seq = [0]
    finished = False
    sends: list = []                                       # work handles of this rank's sends (settled per session)

    def deliver(clip: int, res) -> None:
        """One finished clip: to the local writers (rank 0) or on its way to rank 0."""
        if res is None:
            publish(clip, _ST_EMPTY, 0, False)
            return
        n, has_probs = res.frames, res.probs is not None
        if rank == 0:
            rows = res.rows.cpu().numpy() if res.on_device else res.rows
            probs = (res.probs.cpu().numpy() if res.on_device else res.probs) if has_probs else None
            local_done[clip] = (rows, probs)
            publish(clip, _ST_OK, n, has_probs)
            return
        if lw:                                             # this rank's own writer threads take the clip
            rows = res.rows.cpu().numpy() if res.on_device else res.rows
            probs = (res.probs.cpu().numpy() if res.on_device else res.probs) if has_probs else None
            records[clip]["rank"], records[clip]["status"] = rank, "ok"
            writer.submit(clip, rows, probs)
            publish(clip, _ST_OK, n, has_probs)
            return
        if n:                                              # the sends are posted BEFORE the ticket that announces them
            rows = res.rows if nccl else torch.from_numpy(res.rows) if not res.on_device else res.rows.cpu()
            work = _p2p(dist.isend, rows.contiguous(), 0)
            probs = None
            if has_probs:
                probs = res.probs if nccl else torch.from_numpy(res.probs) if not res.on_device else res.probs.cpu()
                work += _p2p(dist.isend, probs.contiguous(), 0)
            waits = [_SettleOnce(w, gather_given_up) for w in work]
            res.pending.extend(waits)                      # before the session (whose buffers the sends read) is reused
            sends.append((waits, rows, probs))             # ... and before this rank leaves; a transfer is waited for ONCE
            sends[:] = [s_ for s_ in sends if not all(o.done for o in s_[0])]
        publish(clip, _ST_OK, n, has_probs)

    rescue_runner: list = []

    def failed(clip: int, e: BaseException) -> None:
        """A clip raised.  When its activations left the encoder's range (CBAS_ERANGE) it is run again, synchronously, through
        the encoder's precision-3 twin (pipeline.range_fallback_for: the reference's fp32 has no range limit); anything else
        - and a second failure - is logged and skipped, as EncodeThread does (workthreads.py:334-336)."""
        try:
            twin = P.range_fallback_for(encoder, e, paths[clip])
            if twin is not None:
                if not rescue_runner:
                    rescue_runner.append(P.ClipRunner(twin, head, temperature, sessions=1))
                deliver(clip, rescue_runner[0].run(paths[clip], None, progress_callback, device_out=devrows and rank != 0 and not lw))
                return
        except Exception as e2:  # noqa: BLE001
            e = e2
        print(f"ERROR during encoding for {paths[clip]} on rank {rank}: {e}")
        publish(clip, _ST_FAILED, 0, False)

    # Clips are pipelined back to back where the results go to HOST memory (rank 0, gloo, one process): clip i+1 is pushed
    # before clip i's tail - its last batches, the tail classification, the copy-out - is waited for, so the GPU never idles
    # between clips (a clip's fixed cost was ~8 ms: 10 % of a 2 048-frame clip).  On RCCL ranks > 0 the rows leave from
    # the session's device buffers instead (two sessions alternate there already).
    pipelined = lw or not (devrows and rank != 0)
    if pipelined and runner.native and len(runner._sessions) < 2:
        runner._sessions.append(None)
    prev = None                                            # (clip, pending result) of the clip before the current one
    # Look-ahead: while clip i is pushed, a helper thread opens clip i+1 and decodes its first pieces into page-locked buffers,
    # so the GPU does not idle through "open the file, walk its index, decode 128 frames" (8-9 ms in a rocprofv3 timeline)
    # between clips.  The next clip is taken from the queue one clip early for that - only while at least one untaken clip
    # per rank is left, so the queue's tail stays as dynamic as it was.
    ahead = None                                           # (clip, Future of a prepared clip) taken from the queue early
    pool = None
    if pipelined and runner.native:
        from concurrent.futures import ThreadPoolExecutor
        pool = ThreadPoolExecutor(max_workers=1, thread_name_prefix="cbas-open-ahead")
    try:
        while True:
            if ahead is not None:
                clip, fut = ahead
                ahead = None
                prepared = fut.result()
            else:
                clip, prepared = queue_.next(), None
            if clip is None:
                break
            taken.add(clip)
            if pool is not None and queue_.remaining() >= world:
                nxt = queue_.next()
                if nxt is not None:
                    taken.add(nxt)
                    ahead = (nxt, pool.submit(runner.prepare, paths[nxt]))
            try:
                if pipelined:
                    cur = runner.submit(paths[clip], None, progress_callback, prepared=prepared)
                else:
                    cur = P._PendingClip(runner.run(paths[clip], None, progress_callback, device_out=True), None)
            except Exception as e:  # noqa: BLE001 - the queue survives a bad file (workthreads.py:334-336)
                failed(clip, e)
                continue
            finally:
                if prev is not None:                       # the clip before: its tail has had a whole clip's time to drain
                    pc, pp = prev
                    prev = None
                    try:
                        deliver(pc, pp.result())
                    except Exception as e:  # noqa: BLE001
                        failed(pc, e)
            if cur.done:                                   # nothing queued behind it (stand-in encoder, device outputs): now
                try:
                    deliver(clip, cur.result())
                except Exception as e:  # noqa: BLE001
                    failed(clip, e)
            else:
                prev = (clip, cur)
        if prev is not None:
            pc, pp = prev
            prev = None
            try:
                deliver(pc, pp.result())
            except Exception as e:  # noqa: BLE001
                failed(pc, e)
        for waits, _r, _p in sends:
            for o in waits:
                o()
        finished = True
    finally:
        if ahead is not None:                              # an error is on its way up with a clip opened ahead
            try:
                ahead[1].result().close()
            except BaseException:  # noqa: BLE001
                pass
        if pool is not None:
            pool.shutdown(wait=True)
        # every clip this rank took from the queue gets exactly one ticket, whatever happened to it: a clip drawn early
        # for look-ahead, or one whose delivery raised outside the per-clip handlers, would otherwise leave rank 0's
        # receiver waiting for a ticket that never comes
        for c in sorted(taken - published):
            try:
                print(f"ERROR during encoding for {paths[c]} on rank {rank}: not delivered (this rank is leaving early)")
                publish(c, _ST_FAILED, 0, False)
            except Exception:  # noqa: BLE001
                pass
        if lw and rank != 0:
            # local_writes: wait for this rank's files, then tell rank 0 how each of its clips ended (before `left`)
            try:
                writer.close()
                mine = {str(c): [r["status"], r["cls_file"], r["csv_file"], r["frames"]] for c, r in enumerate(records) if r["rank"] == rank}
                store.set(f"{prefix}written{rank}", json.dumps(mine))
            except Exception as e:  # noqa: BLE001
                print(f"ERROR closing the writers of rank {rank}: {e}")
        if store is not None:
            hb_stop.set()
            try:
                store.set(f"{prefix}left{rank}", "1")
            except Exception:  # noqa: BLE001
                pass
        if rank == 0:
            if not finished and world == 1:
                stop.set()                                 # an error is on its way up and nobody else can publish: do not wait
            receiver.join()
            writer.close()
            if lw:
                # the other ranks' outcomes: each publishes `written<r>` before `left<r>`; a rank that is gone without it
                # leaves its clips "failed"
                lv = _Liveness(store, prefix, world, dead_after)
                for r in range(1, world):
                    polls = 0
                    while not store.check([f"{prefix}written{r}"]):
                        if lv.left(r) and not store.check([f"{prefix}written{r}"]):
                            break
                        polls += 1
                        if polls % 250 == 0 and lv.silent(r):      # its heartbeat value has not changed for dead_after s of OUR clock
                            break
                        time.sleep(0.002)
                    got = json.loads(store.get(f"{prefix}written{r}").decode()) if store.check([f"{prefix}written{r}"]) else {}
                    for c, rec in enumerate(records):
                        if rec["rank"] == r and rec["status"] == "ok":
                            st_, h5_, csv_, fr_ = got.get(str(c), ["failed", None, None, 0])
                            rec["status"], rec["cls_file"], rec["csv_file"] = st_, h5_, csv_
                            if st_ != "ok":
                                rec["cls_file"] = rec["csv_file"] = None
        runner.close()
        for rr in rescue_runner:
            rr.close()
    if recv_err:
        raise recv_err[0]
    return records if rank == 0 else None


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


class _FrameRange:
    """Frames [start, stop) of a reader, as a reader (``len``, ``get_batch``, ``read_into`` / ``frame_shape`` when the
    underlying reader has them)."""

    def __init__(self, reader, start: int, stop: int):
        self._r, self._a, self._b = reader, int(start), int(stop)
        if hasattr(reader, "frame_shape") and hasattr(reader, "read_into"):
            self.frame_shape = reader.frame_shape
            self.read_into = lambda i, j, out: reader.read_into(self._a + i, self._a + j, out)
            if hasattr(reader, "read_channel_into"):
                self.read_channel_into = lambda i, j, ch, out: reader.read_channel_into(self._a + i, self._a + j, ch, out)

    def __len__(self):
        return self._b - self._a

    def get_batch(self, indices):
        idx = indices if isinstance(indices, range) else list(indices)
        if isinstance(idx, range) and idx.step == 1:
            return self._r.get_batch(range(self._a + idx.start, self._a + idx.stop))
        return self._r.get_batch([self._a + int(i) for i in idx])


def encode_infer_file_sharded(path: str, encoder, head=None, dataset_name: Optional[str] = None,
                              behaviors: Optional[Sequence[str]] = None, temperature: float = 1.0,
                              progress_callback=None) -> Tuple[Optional[str], Optional[str]]:
    """ONE video on all ranks (last sentence): rank r decodes and encodes the contiguous frame range
    ``shard_frames(n, world, r)``, neighbouring ranks exchange the ``seq_len // 2`` CLS rows on either side of each cut
    (``exchange_halo`` - the one real exchange step of the path, 23 KB per cut), every rank classifies its own frames, and
    the rows / probabilities are gathered to rank 0 (``gather_rows``), which writes ``<video>_cls.h5`` and, with a head,
    ``<video>_<dataset>_outputs.csv`` - byte for byte the files ``encode_file`` + ``infer_file`` write on one GPU (frames
    are encoded independently; a window that straddles a cut sees the same rows).  Use it when there are fewer videos than
    GPUs (one 30-minute clip at 10 fps = 18 000 frames: 2 250 per GPU at 8).  Every rank must call it; returns the two
    paths on rank 0 (``None`` for the CSV without a head), (None, None) elsewhere and for a video without frames.
    The reader must allow random access (decord, ``.npy``, Motion-JPEG AVI do; a sequential decoder pipe does not)."""
    from . import pipeline as P
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    if head is not None and (dataset_name is None or behaviors is None):
        raise ValueError("classification needs dataset_name and behaviors (infer_file's arguments)")
    D = encoder.config.hidden_size
    dev = encoder.device if _on_gpu(encoder) else torch.device("cpu")

    def agree(err: Optional[BaseException], what: str, value: int = 0) -> None:
        """Every rank arrives here whether or not its own step failed; ALL ranks leave by raising when ANY failed (or when
        they disagree on ``value``), so no rank goes on into a collective the others have abandoned: a clip that cannot be
        read in one rank's frame range is skipped on every rank together (the caller logs it and takes the next video, as
        EncodeThread does: workthreads.py:334-336) instead of hanging the job."""
        if world > 1:
            t = torch.tensor([0 if err is not None else 1, value, -value], dtype=torch.int64,
                             device="cpu" if dist.get_backend() == "gloo" else dev)
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
            ok, vmin, vmax = bool(t[0].item()), int(t[1].item()), -int(t[2].item())
        else:
            ok, vmin, vmax = err is None, value, value
        if err is not None:
            raise err
        if not ok:
            raise RuntimeError(f"{path}: {what} failed on another rank; the clip is skipped on every rank")
        if vmin != vmax:
            raise RuntimeError(f"{path}: the ranks do not see the same video ({vmin} .. {vmax} frames)")

    reader, n, err = None, 0, None
    try:
        reader = P.open_video(path)
        if getattr(reader, "decodes_ahead", False) and world > 1:
            raise RuntimeError(f"{path}: its frame source reads sequentially; a clip can only be split over ranks with a "
                               "random-access reader")
        n = len(reader)
        if head is not None and hasattr(head, "to"):
            head.to(dev)
    except Exception as e:  # noqa: BLE001 - agreed on below
        err = e
    try:
        agree(err, "opening the video", n)
        if n == 0:
            if rank == 0:
                print(f"Warning: Video {path} contains no frames. Skipping.")
            return None, None
        a, b = shard_frames(n, world, rank)
        runner = P.ClipRunner(encoder, None)
        try:
            rows = None
            try:
                if b > a:
                    res = runner.run(path, _FrameRange(reader, a, b), progress_callback if rank == 0 else None, device_out=True)
                    rows = res.rows if isinstance(res.rows, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(res.rows))
                if rows is None:
                    rows = torch.empty((0, D), dtype=torch.float16, device=dev)
            except Exception as e:  # noqa: BLE001
                err = e
            agree(err, f"encoding frames of a rank's range")
            probs = None
            if head is not None:
                # classify_sharded, with the agreement between its exchange step and the rank-local head call
                half = head.seq_len // 2
                left, right = exchange_halo(rows, half)
                try:
                    buf = torch.cat([left, rows, right]).contiguous()
                    pr = torch.empty((buf.shape[0], head.out_features), dtype=torch.float32, device=buf.device)
                    if rows.shape[0]:
                        head.infer_range_into(buf, buf.shape[0], left.shape[0], rows.shape[0], pr, temperature)
                    probs = pr[left.shape[0]:left.shape[0] + rows.shape[0]].contiguous()
                except Exception as e:  # noqa: BLE001
                    err = e
                agree(err, "classifying a rank's range")
            g_rows = gather_rows([rows.contiguous()], dst=0)
            g_probs = gather_rows([probs], dst=0) if probs is not None else None
            if rows.is_cuda:
                torch.cuda.synchronize(rows.device)        # the transfers read the session's buffers: done before it closes
            if rank != 0:
                return None, None
            all_rows = torch.cat([t for per in g_rows for t in per]).cpu().numpy()
            assert all_rows.shape == (n, D), (all_rows.shape, n)
            cls_path = P.write_cls_file(path, all_rows, P.file_attrs(encoder))
            print(f"Successfully encoded {os.path.basename(path)} to {os.path.basename(cls_path)}")
            csv_path = None
            if g_probs is not None:
                all_probs = torch.cat([t for per in g_probs for t in per]).cpu().numpy()
                csv_path = cls_path.replace("_cls.h5", f"_{dataset_name}_outputs.csv")
                P.write_probs_csv(csv_path, all_probs, list(behaviors))
            return cls_path, csv_path
        finally:
            runner.close()
    finally:
        if reader is not None and hasattr(reader, "close"):
            reader.close()


def _on_gpu(encoder) -> bool:
    dev = getattr(encoder, "device", None)
    return dev is not None and getattr(dev, "type", "cpu") == "cuda"

