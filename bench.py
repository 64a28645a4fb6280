#!/usr/bin/env python3
"""Headline benchmark: frames/s of the streamed encode + classify path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1]): DINOv3 ViT-B/16, synthetic 224x224 RGB uint8 clip, batch 64, chunked
encode -> fp16 CLS -> sliding-window BiLSTM head (C=9, seq_len 31).  One *step* = one 64-frame batch through
the encoder, plus the head over every frame whose 31-frame window has become complete (classified in groups,
the tail inside the timed region).  Weights are synthetic (seeded, counter-based): the real checkpoints are
gated and there is no network.  With N > 1 each rank streams its own clip (weak scaling, no data-path
collective) and the output rows are gathered to rank 0 over RCCL inside the timed region.

Passes over the same K steps:
  1. `value`: SURVEY section 8(d)'s metric definition - uint8 RGB in pinned host memory -> cbas_fused_push_u8_host
     (PCIe H2D on the copy stream) -> encoder -> head -> fp16 CLS rows + fp32 probabilities back in host memory
     (with N > 1: in rank 0's host memory, after the RCCL gather).  PCIe inclusive;
  2. `hbm_resident`: the same K steps with the frames already in HBM when the clock starts and the results left in
     HBM (the looser figure; it was `value` until round 2), bit-identical outputs;
  3. per-kernel HIP-event timing for `roofline` (one batch in flight);
  4. `files_path` (--files F, default 2 clips per rank): the PRODUCT path - cbas_amd.dist.encode_files on synthetic
     `.npy` clips: shared clip queue, decode-ahead into page-locked buffers, fused encode + classify, gather to rank 0,
     `_cls.h5` + `_outputs.csv` written by rank 0's writer threads; wall time from the first clip to the last file.
`gates` = the correctness gates of section 8(d) evaluated in this very process against the fixtures made from the
reference (tests/golden): CLS relative error on the golden frames, head label mismatches on golden rows.

Prints ONE JSON line on rank 0 (see README/DESIGN.md for the fields).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

# RCCL shares device buffers between the ranks' processes through dmabuf IPC handles on this pool's driver; the HSA runtime reads
# the switch when it starts, i.e. before the first device call below (cbas_amd.dist sets it too, but later than set_device)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import numpy as np  # noqa: E402
import torch  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)

from cbas_amd import config as C  # noqa: E402
from cbas_amd import dist as cdist  # noqa: E402
from cbas_amd import weights as W  # noqa: E402
from cbas_amd import synth  # noqa: E402

METRIC = "frames/sec DINOv3-B/16 224px encode+LSTM classify, 1/2/4/8 MI355X"     # BASELINE.json "metric", verbatim
MFMA_F16_PEAK_TFLOPS = 2500.0      # dense fp16/bf16 MFMA peak, /opt/skills/guides/MI355X_MICROARCH.md
MFMA_FP8_PEAK_TFLOPS = 5000.0      # dense fp8 (block-scaled) MFMA peak, same guide
MFMA_F32_PEAK_TFLOPS = 157.3       # f32-input MFMA (v_mfma_f32_16x16x4_f32) = the fp32 vector peak, same guide
BEHAVIORS = 9
SEQ_LEN = 31
MODEL_LABEL = {"vits16": "DINOv3 ViT-S/16", "vitb16": "DINOv3 ViT-B/16", "vitl16": "DINOv3 ViT-L/16", "tiny": "tiny test ViT",
               "dinov2regb14": "DINOv2-with-registers ViT-B/14 (CBAS's default encoder family)", "dinov2regtiny": "tiny DINOv2-with-registers"}


def cpu_baseline(model: str, hw: int, frames: int, batch: int) -> dict:
    """The oracle (CPU float32 restatement of the reference path, torch CPU ops - what the reference's own CPU
    path runs on) timed on this box's host cores, on a bounded sample of the same workload.  Checker only: never
    the product path."""
    from oracle import pipeline_oracle as PO
    from oracle import vit_oracle_torch as VT
    # use the cores this process may actually run on: the GPU box gives a CPU *share* through the cgroup quota (16 cores
    # per GPU) while still showing every core of the host - 64 threads on a 16-core quota ran 2.8x slower than 16
    try:
        threads = len(os.sched_getaffinity(0))
    except AttributeError:
        threads = os.cpu_count() or 1
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            threads = min(threads, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    threads = max(1, min(threads, int(os.environ.get("CBAS_CPU_BASELINE_THREADS", "64"))))
    old_threads = torch.get_num_threads()
    torch.set_num_threads(threads)
    cfg = C.NAMED_VIT[model]
    hcfg = C.HeadConfig(in_features=cfg.hidden_size, out_features=BEHAVIORS, seq_len=SEQ_LEN)
    enc_w = VT.to_torch(W.synth_encoder_weights(cfg, 1234))
    head_w = W.synth_head_weights(hcfg, 4321)
    fr = synth.noise_frames(0, frames, hw, hw)

    def run(f):
        cls32 = VT.encode_frames(f, enc_w, cfg, batch)
        return PO.classify_cls(cls32.astype(np.float16), head_w, SEQ_LEN, 1.0)
    run(fr[:batch])                                   # warm-up batch
    t0 = time.perf_counter()
    run(fr)
    dt = time.perf_counter() - t0
    torch.set_num_threads(old_threads)
    return {"value": round(frames / dt, 3), "unit": "frames/s", "cores": int(threads), "kind": "port",
            "sample": f"{frames} frames of the same workload ({model} {hw}x{hw}, batch {batch}, fp32 torch-CPU "
                      f"restatement of the encoder (oracle/vit_oracle_torch.py) + numpy LSTM head), {dt:.1f} s wall"}


_FRAME_CACHE: dict = {}


def _frames(kind: str, seed: int, n: int, hw: int, first: int = 0):
    """Synthetic fixture frames, generated once per process (the label-exact leg reuses what the default leg made)."""
    key = (kind, seed, n, hw, first)
    if key not in _FRAME_CACHE:
        mk = synth.noise_frames if kind == "noise" else synth.cage_frames
        _FRAME_CACHE[key] = mk(seed, n, hw, hw, first=first)
    return _FRAME_CACHE[key]


def gates(enc, head, model: str, hw: int, precision: int) -> dict:
    """SURVEY section 8(d) 'correctness gates reported with every number', against fixtures generated from the
    reference (tests/golden/make_goldens.py): max per-frame ||CLS - ref||2 / ||ref||2 on the golden frames of this
    model/resolution, and argmax mismatches of the head on the reference's 700-frame infer_file golden."""
    gd = os.path.join(HERE, "tests", "golden")
    out = {"cls_tol": 5e-6 if precision >= 3 else 1e-3 if precision < 2 else None}
    name = {("vitb16", 224): "vitb16_224_noise", ("vitb16", 256): "vitb16_256", ("vits16", 224): "vits16_224",
            ("vitl16", 224): "vitl16_224", ("vitl16", 518): "vitl16_518"}.get((model, hw))
    path = os.path.join(gd, f"{name}.npz") if name else None
    if path and os.path.exists(path):
        g = np.load(path)
        fr = _frames("noise" if str(g["kind"]) == "noise" else "cage", int(g["frame_seed"]), int(g["n"]), hw)
        _, c32 = enc.encode_u8(torch.from_numpy(fr).to(enc.device))
        c32 = c32.cpu().numpy().astype(np.float64)
        ref = g["cls"].astype(np.float64)
        rel = np.linalg.norm(c32 - ref, axis=1) / np.linalg.norm(ref, axis=1)
        out.update(cls_rel_err_max=float(f"{rel.max():.3e}"), cls_frames=int(len(rel)), cls_fixture=f"{name}.npz")
    else:
        out.update(cls_rel_err_max=None, cls_frames=0, cls_fixture=None)
    ip = os.path.join(gd, "infer_file.npz")
    if os.path.exists(ip) and head.in_features == 768 and head.out_features == BEHAVIORS:
        g = np.load(ip)
        cls = torch.from_numpy(synth.cls_walk(100 + 700, 700, 768)).to(enc.device)
        pr = head.infer_clip(cls, float(g["temp_700"])).cpu().numpy()
        ref = g["probs_700"]
        out.update(head_label_mismatches=int((pr.argmax(1) != ref.argmax(1)).sum()), head_frames=700,
                   head_prob_err_max=float(f"{np.abs(pr - ref).max():.3e}"), head_fixture="infer_file.npz[probs_700]")
    # end to end on the headline model: frames -> this encoder -> f16 rows -> this head, against the reference's own
    # DinoEncoder wrapper + infer_file (tests/golden/e2e_vitb16.npz, 256 frames)
    ep = os.path.join(gd, "e2e_vitb16.npz")
    if os.path.exists(ep) and (model, hw) == ("vitb16", 224) and head.in_features == 768 and head.out_features == BEHAVIORS:
        g = np.load(ep)
        fr = _frames("cage", int(g["frame_seed"]), int(g["n"]), hw)
        c16, _ = enc.encode_u8(torch.from_numpy(fr).to(enc.device), want_f32=False)
        pr = head.infer_clip(c16, 1.0).cpu().numpy().astype(np.float64)
        ref = g["probs"].astype(np.float64)
        srt = np.sort(ref, axis=1)
        margin = srt[:, -1] - srt[:, -2]
        flips = np.nonzero(pr.argmax(1) != ref.argmax(1))[0]
        c = c16.float().cpu().numpy().astype(np.float64)
        rc = g["cls"].astype(np.float64)
        rel = np.linalg.norm(c - rc, axis=1) / np.linalg.norm(rc, axis=1)
        # the reference's OWN labels under its other legitimate executions (1 frame per encoder call; MKL on AVX2): a frame the
        # reference labels both ways is matched by either label (tests/test_gpu_fp32.py::_strict_gate)
        import glob
        lab, unexplained = pr.argmax(1), pr.argmax(1) != ref.argmax(1)
        for vp in sorted(glob.glob(os.path.join(gd, "e2e_vitb16_variants*.npz"))):
            v = np.load(vp)
            for k in v.files:
                if k.startswith("labels_"):
                    unexplained &= lab != v[k]
        out["e2e"] = {"fixture": "e2e_vitb16.npz", "frames": int(len(ref)), "label_mismatches": int(len(flips)),
                      "label_mismatches_vs_every_reference_variant": int(unexplained.sum()),
                      "largest_reference_margin_at_a_mismatch": float(f"{margin[flips].max():.3e}") if len(flips) else None,
                      "reference_frames_with_margin_under_1e-2": int((margin < 1e-2).sum()),
                      "prob_err_max": float(f"{np.abs(pr - ref).max():.3e}"),
                      "cls_f16_rel_err_max": float(f"{rel.max():.3e}")}
    # the same on the long clip (2 048 frames, 122 behaviour transitions): the flip RATE of this arithmetic and where the
    # flips sit on the reference's own top-2 margin scale
    lp = os.path.join(gd, "e2e_vitb16_long.npz")
    if os.path.exists(lp) and (model, hw) == ("vitb16", 224) and head.in_features == 768 and head.out_features == BEHAVIORS:
        g = np.load(lp)
        n = int(g["n"])
        c16 = torch.empty((n, 768), dtype=torch.float16, device=enc.device)
        for i in range(0, n, 512):
            fr = _frames("cage", int(g["frame_seed"]), min(512, n - i), hw, first=i)
            c16[i:i + len(fr)] = enc.encode_u8(torch.from_numpy(fr).to(enc.device), want_f32=False)[0]
        pr = head.infer_clip(c16, 1.0).cpu().numpy().astype(np.float64)
        ref = g["probs"].astype(np.float64)
        srt = np.sort(ref, axis=1)
        margin = srt[:, -1] - srt[:, -2]
        flips = np.nonzero(pr.argmax(1) != ref.argmax(1))[0]
        edges = [0.0, 1e-4, 1e-3, 3e-3, 1e-2, 3e-2, 1e-1, 1.0]
        out["e2e_long"] = {"fixture": "e2e_vitb16_long.npz", "frames": n,
                           "reference_label_transitions": int((g["labels"][1:] != g["labels"][:-1]).sum()),
                           "label_mismatches": int(len(flips)), "flip_rate": float(f"{len(flips) / n:.3e}"),
                           "largest_reference_margin_at_a_mismatch": float(f"{margin[flips].max():.3e}") if len(flips) else None,
                           "margin_bin_edges": edges,
                           "reference_frames_per_margin_bin": np.histogram(margin, edges)[0].tolist(),
                           "mismatches_per_margin_bin": np.histogram(margin[flips], edges)[0].tolist(),
                           "prob_err_max": float(f"{np.abs(pr - ref).max():.3e}"),
                           "fp16_elements_differing_pct": float(f"{(c16.cpu().numpy() != g['cls_f16']).mean() * 100:.3f}")}
    return out


def labels_identical(g: dict):
    """north_star's "identical argmax labels", from the gates of one mode: every label of both end-to-end fixtures equals the
    reference's - where the reference labels a frame both ways under its own execution variants, either of ITS labels.  None
    when the fixtures were not evaluated (another model / frame size)."""
    if not g or "e2e" not in g or "e2e_long" not in g:
        return None
    return bool(g["e2e"]["label_mismatches_vs_every_reference_variant"] == 0 and g["e2e_long"]["label_mismatches"] == 0)


def files_pass(args, enc, head, rank: int, world: int, device) -> dict:
    """The product path on files: every rank writes `--files` synthetic `.npy` clips (uint8 (n,H,W,3), what decord hands
    encode_file), then all ranks drain the whole list with cbas_amd.dist.encode_files - shared clip queue, decode-ahead
    into page-locked buffers, fused encode + classify, rows gathered to rank 0, `_cls.h5` + `_outputs.csv` written by rank
    0's writer threads.  One untimed pass (buffers, sessions, communicator), one timed pass, barrier to barrier."""
    import shutil
    import tempfile
    from cbas_amd import pipeline as P
    names = [f"b{i}" for i in range(BEHAVIORS)]
    root = args.files_dir
    if rank == 0 and root is None:
        root = tempfile.mkdtemp(prefix="cbas_bench_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    if world > 1:
        box = [root]
        torch.distributed.broadcast_object_list(box, src=0, device=device if torch.distributed.get_backend() == "nccl" else None)
        root = box[0]
    os.makedirs(root, exist_ok=True)
    n, hw = args.clip_frames, args.hw
    gen = torch.Generator(device=device)
    avi = args.clip_format == "avi"
    jpeg_bytes = 0
    for j in range(args.files):
        gen.manual_seed(7000 + rank * 100 + j)
        if avi:
            # camera-like pictures (a smooth field + sensor noise, grey on all three channels), coded 4:2:0 at quality 85
            from cbas_amd.framesource import write_mjpeg_avi
            low = torch.rand((n, 1, hw // 16, hw // 16), device=device, generator=gen)
            g = torch.nn.functional.interpolate(low, size=(hw, hw), mode="bicubic", align_corners=False) * 200.0 + 28.0
            g = g + torch.randn((n, 1, hw, hw), device=device, generator=gen) * 4.0
            g8 = g.clamp_(0, 255).to(torch.uint8)[:, 0]
            fr = g8[..., None].expand(n, hw, hw, 3).contiguous().cpu().numpy()
            path = os.path.join(root, f"clip_r{rank:02d}_{j:02d}.avi")
            write_mjpeg_avi(path, fr, quality=85, subsampling=2)
            jpeg_bytes += os.path.getsize(path)
        else:
            fr = torch.randint(0, 256, (n, hw, hw, 3), dtype=torch.uint8, device=device, generator=gen).cpu().numpy()
            np.save(os.path.join(root, f"clip_r{rank:02d}_{j:02d}.npy"), fr)
        del fr
    cdist.barrier()
    paths = sorted(os.path.join(root, f) for f in os.listdir(root) if f.endswith(".avi" if avi else ".npy"))
    P.set_project_stamp("bench/synthetic-" + args.model)
    try:
        def once():
            cdist.barrier()
            torch.cuda.synchronize(device)
            t0 = time.perf_counter()
            recs = cdist.encode_files(paths, enc, head=head, dataset_name="bench", behaviors=names, temperature=1.0)
            cdist.barrier()
            return cdist.max_over_ranks(time.perf_counter() - t0, device), recs
        # encode_files reports each clip as the reference does ("Successfully encoded ..." on stdout): stdout carries only
        # the ONE JSON line of this script, so those go to stderr here
        import contextlib
        with contextlib.redirect_stdout(sys.stderr):
            once()
            dt, recs = once()
    finally:
        P.set_project_stamp(None)
    out = None
    if rank == 0:
        ok = [r for r in recs if r["status"] == "ok" and r["cls_file"] and r["csv_file"]]
        frames = sum(r["frames"] for r in ok)
        per_rank = [sum(1 for r in recs if r["rank"] == k) for k in range(world)]
        out = {"value": round(frames / dt, 2), "unit": "frames/s", "clips": len(paths), "clips_ok": len(ok),
               "frames_per_clip": n, "seconds": round(dt, 4), "clips_per_rank": per_rank,
               "bytes_written": int(sum(os.path.getsize(r["cls_file"]) + os.path.getsize(r["csv_file"]) for r in ok)),
               "what": "cbas_amd.dist.encode_files over synthetic " +
                       ("Motion-JPEG AVI clips (4:2:0, quality 85, %.1f KB per frame; decoded by cbas_mjpeg_decode on host "
                        "threads -> green planes in the page-locked ring" % (jpeg_bytes / 1024.0 / max(1, n * args.files)) if avi else
                        ".npy clips (page cache -> page-locked ring") +
                       " -> HBM -> fused encode + classify -> gather to rank 0 -> _cls.h5 + _outputs.csv on rank 0's writer "
                       "threads); barrier to barrier, second pass"}
    cdist.barrier()
    if rank == 0 and args.files_dir is None:
        shutil.rmtree(root, ignore_errors=True)
    return out


def _null_line(args, n_gpus: int, ranks_seen, error: str) -> str:
    return json.dumps({"metric": METRIC, "value": None, "unit": "frames/s", "n_gpus": n_gpus, "ranks_seen": ranks_seen,
                       "steps": args.steps, "warmup": args.warmup, "ms_per_step": None, "higher_is_better": True, "scaling": "weak",
                       "vs_baseline": None, "error": error})


def launch_ranks(args) -> int:
    """`python bench.py --gpus N` with N > 1 and no launcher around it: start the N ranks HERE, as children
    (`python -m torch.distributed.run --nproc-per-node N bench.py ...`), before this process has made any GPU call (a process
    that touched the GPU must never be replaced, so nothing is re-exec'ed), relay rank 0's single JSON line and return the
    children's worst exit code.  With fewer than N devices visible (RCCL refuses two ranks on one device) the line carries
    value null and the reason, and the exit code is 2: an N-rank request never degrades into a one-GPU measurement.
    CBAS_DIST_BACKEND=gloo rehearses the N-rank control flow on fewer devices (ranks share them; a rehearsal, not a number)."""
    import socket
    import subprocess
    backend = os.environ.get("CBAS_DIST_BACKEND", "nccl")
    n_dev = torch.cuda.device_count()                       # counting devices does not initialise the GPU on this image
    if backend == "nccl" and n_dev < args.gpus:
        print(_null_line(args, args.gpus, 0, f"--gpus {args.gpus} asked for {args.gpus} ranks over RCCL but only {n_dev} device(s) are "
                         "visible (RCCL refuses two ranks on one device); nothing was measured. CBAS_DIST_BACKEND=gloo rehearses "
                         "the control flow on fewer devices"), flush=True)
        return 2
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__), *sys.argv[1:]]
    env = dict(os.environ, CBAS_BENCH_LAUNCHED="1")
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // args.gpus)))
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env, text=True)
    line = None
    for ln in proc.stdout:                                   # every rank's descriptor 1 points at stderr except rank 0's record
        ln = ln.strip()
        if ln.startswith("{") and '"metric"' in ln:
            line = ln
    rc = proc.wait()
    if line is None:
        line = _null_line(args, args.gpus, None, f"the {args.gpus}-rank run printed no record (torch.distributed.run exit code {rc})")
        rc = rc or 5
    print(line, flush=True)
    return rc


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=157)        # 157 x 64 = 10 048 frames: the 10k-frame clip
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--model", default="vitb16")
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--hw", type=int, default=224)
    ap.add_argument("--precision", type=int, default=0)
    ap.add_argument("--cpu-frames", type=int, default=512)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--no-host-path", action="store_true")
    ap.add_argument("--no-gates", action="store_true")
    ap.add_argument("--no-label-exact", action="store_true",
                    help="skip the secondary leg that runs the same clip in precision 4 (the label-exact mode) and reports its "
                         "frames/s and gates beside the default mode's")
    ap.add_argument("--lanes", type=int, default=2, help="batches in flight per GPU (compute lanes of the encoder)")
    ap.add_argument("--files", type=int, default=2, help="clips per rank of the files_path pass (0: skip it)")
    ap.add_argument("--clip-frames", type=int, default=4096, help="frames per clip of the files_path pass")
    ap.add_argument("--clip-format", choices=("npy", "avi"), default="npy",
                    help="files_path clips: raw uint8 frames (.npy) or Motion-JPEG AVI (real decoder work in the loop)")
    ap.add_argument("--files-dir", default=None, help="where the synthetic clips go (default: a temp dir under /dev/shm)")
    ap.add_argument("--host-input", choices=("green", "rgb"), default="green",
                    help="`value` pass: how the pinned RGB frames reach the device.  green (default, r4): a decode-ahead thread "
                         "keeps channel 1 while it fills page-locked ring pieces (cbas_pick_channel_u8), as the file paths do - "
                         "50 176 bytes per 224x224 frame cross PCIe (SURVEY section 8(d)); rgb: the r3 form - the RGB bytes are "
                         "DMA'd as they are (150 528 per frame) and the device picks the channel")
    ap.add_argument("--preroll-seconds", type=float, default=1.0,
                    help="untimed steady-state work right before each timed pass, besides the W warm-up steps: a short run "
                         "(the driver's --steps 20 is 58 ms) otherwise starts from an idle device - clocks down, power "
                         "management settling - and reads several % under a 157-step run of the same binary")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:     # no launcher around us: be the launcher (before any GPU call)
        sys.exit(launch_ranks(args))

    # stdout carries exactly ONE line: the JSON record.  Libraries print banners of their own at the file-descriptor level
    # (RCCL's version block when the communicator is created, gloo's "[Gloo] Rank ..." lines, the reference-style
    # "Successfully encoded ..." messages of the file path), so descriptor 1 points at stderr for the whole run - on every
    # rank - and the record is written to the saved descriptor at the end.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    def emit(line: str) -> None:
        sys.stdout.flush()
        os.write(real_stdout, (line + "\n").encode())

    # Whole-run watchdog: the N > 1 passes contain RCCL collectives and point-to-point transfers that no multi-GPU node has
    # run yet.  If the run has not finished in time, rank 0 still writes a line that says so (value null) and every rank
    # exits non-zero, instead of hanging until the caller's own limit with nothing on stdout.
    import threading
    run_limit = float(os.environ.get("CBAS_BENCH_TIMEOUT", str(900 + 0.05 * args.steps)))     # long --steps runs get their time
    finished = threading.Event()

    def _run_watchdog():
        if finished.wait(run_limit):
            return
        if int(os.environ.get("RANK", "0")) == 0:
            emit(json.dumps({"metric": METRIC, "value": None, "unit": "frames/s", "n_gpus": int(os.environ.get("WORLD_SIZE", "1")),
                             "steps": args.steps, "warmup": args.warmup, "higher_is_better": True,
                             "error": f"bench.py did not finish within {run_limit:.0f} s (CBAS_BENCH_TIMEOUT); exit code 4"}))
        os._exit(4)
    threading.Thread(target=_run_watchdog, name="cbas-bench-watchdog", daemon=True).start()

    # RCCL ("nccl") on a real multi-GPU node.  CBAS_DIST_BACKEND=gloo rehearses the multi-rank control
    # flow on a box with fewer GPUs than ranks (ranks then share devices; the gather goes through host
    # memory) - a rehearsal, not a measurement.
    backend = os.environ.get("CBAS_DIST_BACKEND", "nccl")
    if backend == "nccl" and int(os.environ.get("WORLD_SIZE", "1")) > 1:
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
    rank, world, local = cdist.init_from_env(backend)
    if world != args.gpus:                                   # a launcher started another number of ranks than the command line names
        if rank == 0:
            emit(_null_line(args, args.gpus, world, f"--gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks; nothing was measured"))
        finished.set()
        sys.exit(2)
    local = local % max(1, torch.cuda.device_count())
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    ranks_seen = world
    if world > 1:                                            # what the process group itself says after its first collective
        import torch.distributed as tdist
        cdist.barrier()
        ranks_seen = int(tdist.get_world_size())

    from cbas_amd.encoder import DinoEncoder
    from cbas_amd.head import ClassifierLSTMDeltas
    from cbas_amd.stream import ClipStream

    cfg = C.NAMED_VIT[args.model]
    hcfg = C.HeadConfig(in_features=cfg.hidden_size, out_features=BEHAVIORS, seq_len=SEQ_LEN)
    enc = DinoEncoder.from_weights(cfg, W.synth_encoder_weights(cfg, 1234), device, max_batch=args.batch,
                                   max_frame=(args.hw, args.hw), precision=args.precision)
    head = ClassifierLSTMDeltas(cfg.hidden_size, BEHAVIORS, seq_len=SEQ_LEN)
    head.load_state_dict(W.synth_head_weights(hcfg, 4321))
    head.to(device)
    enc.set_lanes(args.lanes)

    B, K, Wm = args.batch, args.steps, args.warmup
    # synthetic clip resident in HBM before the timed region: uniform uint8 RGB, decord layout (n,H,W,3)
    n_res = min(K, 160) * B                      # ~10k frames resident; longer runs wrap around
    gen = torch.Generator(device=device)
    gen.manual_seed(1000 + rank)
    clip = torch.randint(0, 256, (n_res, args.hw, args.hw, 3), dtype=torch.uint8, device=device, generator=gen)
    stream = ClipStream(enc, head, capacity=max(K, Wm) * B, classify_every=1024)
    # the same clip in pinned host memory for the host_path pass (decord hands encode_file host arrays)
    clip_host = None
    if not args.no_host_path:
        clip_host_t = torch.empty((n_res, args.hw, args.hw, 3), dtype=torch.uint8).pin_memory()
        clip_host_t.copy_(clip)
        clip_host = clip_host_t.numpy()

    def run(steps: int):
        stream.reset()
        for s in range(steps):
            o = (s * B) % n_res
            stream.push_u8(clip[o:o + B])
        return stream.finish()

    class _PinnedClip:
        """The pinned RGB clip as a frame source (what decord hands encode_file, already in page-locked memory); wraps
        around after n_res frames like run() does."""
        frame_shape = (args.hw, args.hw, 3)

        def __init__(self, n):
            self.n = n

        def __len__(self):
            return self.n

        def _spans(self, a, b):
            while a < b:
                o = a % n_res
                m = min(b - a, n_res - o)
                yield a, o, m
                a += m

        def read_channel_into(self, a, b, ch, out):
            from cbas_amd import pipeline as P
            for a0, o, m in self._spans(a, b):
                P.pick_channel(clip_host[o:o + m], ch, out[a0 - a:a0 - a + m])

        def read_into(self, a, b, out):
            for a0, o, m in self._spans(a, b):
                np.copyto(out[a0 - a:a0 - a + m], clip_host[o:o + m])

        def get_batch(self, idx):
            return clip_host[np.asarray(list(idx)) % n_res]

    def push_host_frames(steps: int):
        """The frames of `steps` 64-frame batches from pinned host memory into the session; returns a closer to call once
        every host -> HBM copy has completed (the page-locked ring goes back to its pool)."""
        if args.host_input == "rgb":
            for s_ in range(steps):
                o = (s_ * B) % n_res
                stream.push_host(clip_host[o:o + B])
            return lambda: None
        from collections import deque
        from cbas_amd import pipeline as P, _lib as L
        piece = B if (B > P.PIECE and P.CHUNK_SIZE % B == 0) else P.PIECE        # whole batches per ring piece
        chunks = P._chunks(_PinnedClip(steps * B), steps * B, pinned=True, piece=piece)
        held, sub = deque(), 0
        try:
            for _i, _end, frames in chunks:
                stream.push_host(frames)                    # (n, H, W) green planes in a page-locked ring piece
                sub += -(-frames.shape[0] // B)
                held.append((frames, sub + L.ENC_SLOTS))    # a slot's copy is done when the slot is submitted to again
                while held and held[0][1] <= sub:
                    chunks.release(held.popleft()[0])
        except BaseException:
            chunks.close()
            raise
        return chunks.close

    def run_host(steps: int):
        stream.reset()
        done = push_host_frames(steps)
        try:
            return stream.finish_host()           # numpy arrays in host memory; every copy has completed
        finally:
            done()

    def gather(cls16, probs):
        if world > 1:
            cdist.gather_rows([cls16], dst=0)
            cdist.gather_rows([probs], dst=0)

    # warm-up (also builds the RCCL communicator and the rope table outside the timed region)
    t_w = time.perf_counter()
    c16, pr = run(max(Wm, 1))
    gather(c16, pr)
    torch.cuda.synchronize(device)
    # upper bound of a step (first launches included), the same number on every rank: the pre-roll below contains
    # collectives when N > 1, so its trip count must not depend on a rank's own clock
    est_step = cdist.max_over_ranks(max(1e-4, (time.perf_counter() - t_w) / max(Wm, 1)), device)
    cap_steps = max(K, Wm)                                                  # the session's row capacity
    preroll_steps = 0 if args.preroll_seconds <= 0 else max(1, min(cap_steps, int(args.preroll_seconds / est_step)))
    preroll_iters = 0 if not preroll_steps else max(1, min(64, int(np.ceil(args.preroll_seconds / (preroll_steps * est_step)))))

    def preroll(fn):
        """Untimed: the same work back to back for roughly preroll_seconds, then straight into the timed pass."""
        for _ in range(preroll_iters):
            fn(preroll_steps)
        torch.cuda.synchronize(device)

    def timed(pre: bool = True):
        if pre:
            preroll(run)
        cdist.barrier()
        torch.cuda.synchronize(device)
        t0 = time.perf_counter()
        c16, pr = run(K)
        gather(c16, pr)
        torch.cuda.synchronize(device)
        cdist.barrier()
        return cdist.max_over_ranks(time.perf_counter() - t0, device)

    def run_host_gathered(steps: int):
        """The section 8(d) pass: pinned host frames in; rows in (rank 0's) host memory out.  With N > 1 the other ranks'
        rows go HBM -> xGMI -> rank 0's HBM -> rank 0's host memory (no host hop on the sending side)."""
        if world == 1:
            return run_host(steps)
        stream.reset()
        done = push_host_frames(steps)
        try:
            if rank == 0:
                c16h, prh = stream.finish_host()
                g16 = cdist.gather_rows([torch.empty((0, cfg.hidden_size), dtype=torch.float16, device=device)], dst=0)
                gpr = cdist.gather_rows([torch.empty((0, BEHAVIORS), dtype=torch.float32, device=device)], dst=0)
                hosted = [(t[0].cpu(), u[0].cpu()) for t, u in zip(g16[1:], gpr[1:])]       # rank 0 now holds every row
                assert all(t.shape[0] == steps * B for t, _ in hosted)
                return c16h, prh
            c16, pr = stream.finish()
            cdist.gather_rows([c16], dst=0)
            cdist.gather_rows([pr], dst=0)
            torch.cuda.synchronize(device)            # the copies out of the page-locked ring have completed
            return None, None
        finally:
            done()

    def timed_host():
        preroll(run_host_gathered)
        cdist.barrier()
        torch.cuda.synchronize(device)
        t0 = time.perf_counter()
        c16h, prh = run_host_gathered(K)
        torch.cuda.synchronize(device)
        cdist.barrier()
        return cdist.max_over_ranks(time.perf_counter() - t0, device), c16h, prh

    # pass 2 of the docstring first (it also serves as extra warm-up): frames resident in HBM, results left there
    dt = timed()
    # pass 1: the timed region proper (EXACTLY K steps, nothing instrumented): pinned host memory -> host memory -> value
    dt_host, host_equal, dt_r3 = None, None, None
    if clip_host is not None:
        run_host_gathered(max(Wm, 1))
        # the page-locked RESULT buffers of a K-step clip exist before the clock starts, like the page-locked frames do
        # (torch's caching host allocator hands them out again in finish_host; a first hipHostMalloc of 2 MB costs ~0.6 ms,
        # which the W-step warm-up - a shorter clip, a smaller block - does not pay for)
        warm = [torch.empty((K * B, cfg.hidden_size), dtype=torch.float16, pin_memory=True),
                torch.empty((K * B, BEHAVIORS), dtype=torch.float32, pin_memory=True)]
        del warm
        dt_host, c16h, prh = timed_host()
        # the SAME pass as rounds 1-3 defined `value`: RGB bytes over PCIe (green picked on the device), no untimed
        # pre-roll before the clock - so that a reader can separate kernel gains from changes of the metric's definition
        if world == 1 and args.host_input == "green":
            args.host_input = "rgb"
            try:
                run_host_gathered(max(Wm, 1))
                torch.cuda.synchronize(device)
                t0 = time.perf_counter()
                run_host_gathered(K)
                torch.cuda.synchronize(device)
                dt_r3 = time.perf_counter() - t0
            finally:
                args.host_input = "green"
        c16d, prd = run(K)                        # the two passes must agree bit for bit
        torch.cuda.synchronize(device)
        if rank == 0:
            host_equal = bool(np.array_equal(c16d.cpu().numpy().view(np.uint16), c16h.view(np.uint16)) and
                              np.array_equal(prd.cpu().numpy(), prh))
    # pass 2: the same K steps again with every kernel launch bracketed by HIP events on the launch
    # stream -> per-kernel durations for the roofline (the events cost a few % of throughput, which
    # is why they are kept out of pass 1; both wall times are reported)
    # That pass runs one batch at a time (one compute lane): with two batches in flight a kernel's event
    # interval also contains the other lane's kernels, so it would not be that kernel's duration.
    prof, dt_events = {}, None
    if not args.no_kernel_timing:
        torch.cuda.synchronize(device)
        enc.set_lanes(1)
        preroll(run)                              # (outside the profile: its launches must not be counted into K steps' shares)
        enc.profile(True)
        dt_events = timed(pre=False)
        prof = enc.profile_read()
        enc.profile(False)
        enc.set_lanes(args.lanes)

    # The files pass drives RCCL point-to-point transfers from a receiver thread (cbas_amd/dist.py).  No multi-GPU node
    # has run that yet, so it gets a watchdog: if it has not come back in time, the line below is still printed (with the
    # failure stated) and the ranks exit, instead of hanging the run that produced every other number in it.
    files = None
    if args.files > 0:
        import threading
        box: dict = {}

        def _files():
            try:
                torch.cuda.set_device(device)
                box["out"] = files_pass(args, enc, head, rank, world, device)
            except BaseException as e:  # noqa: BLE001
                box["err"] = f"{type(e).__name__}: {e}"
        th = threading.Thread(target=_files, name="cbas-files-pass", daemon=True)
        th.start()
        th.join(float(os.environ.get("CBAS_FILES_PASS_TIMEOUT", "180")))
        if th.is_alive():
            files = {"error": "files_path pass did not finish within its time limit; skipped", "value": None}
            hung = True
        else:
            files = box.get("out") if "err" not in box else {"error": box["err"], "value": None}
            hung = False
    else:
        hung = False

    # A files pass that hung still owns enc / head on its thread: nothing below may touch them, and the run must not
    # read as a success - every rank prints what it has and exits non-zero (no restart, no re-exec).
    if rank != 0:
        if hung:
            sys.stdout.flush()
            os._exit(3)
        finished.set()
        return
    frames_total = K * B * world
    hbm_value = frames_total / dt
    value = frames_total / dt_host if dt_host is not None else hbm_value
    dt_value = dt_host if dt_host is not None else dt
    flops_frame = cfg.flops_per_frame(args.hw, args.hw) + hcfg.flops_per_frame_naive()

    out = {
        "metric": METRIC, "value": round(value, 2), "unit": "frames/s", "n_gpus": world, "ranks_seen": ranks_seen, "steps": K, "warmup": Wm,
        "ms_per_step": round(dt_value / K * 1e3, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": {2: "fp8", 3: "f32", 4: "f32 (GEMM products: 3-term f16 split)"}.get(args.precision, "f16"), "data": "synthetic",
        "config": {"workload": f"{MODEL_LABEL.get(args.model, args.model)} {K * B}-frame synthetic {args.hw}x{args.hw} RGB clip per GPU, "
                               f"batch={B}, chunked encode + BiLSTM head (C={BEHAVIORS}, seq_len={SEQ_LEN})",
                   "input": ("uint8 RGB (n,H,W,3) in pinned host memory -> " +
                             ("channel 1 kept by the decode-ahead thread while it fills page-locked ring pieces (cbas_pick_channel_u8) "
                              "-> cbas_fused_push_u8_host (H2D of the green planes on the copy stream)" if args.host_input == "green" else
                              "cbas_fused_push_u8_host (H2D of the RGB bytes on the copy stream, green picked on the device)") +
                             " -> encoder -> head -> fp16 CLS rows + fp32 probabilities in host memory"
                             + (" of rank 0 (RCCL gather inside the timed region)" if world > 1 else "") +
                             ": SURVEY section 8(d)'s metric definition, PCIe inclusive (HBM-resident figure: hbm_resident)")
                            if dt_host is not None else
                            "uint8 RGB frames resident in HBM before the clock starts; results left in HBM (--no-host-path)",
                   "batch": B, "batches_in_flight": args.lanes, "frames_per_gpu": K * B, "frame": [args.hw, args.hw], "parallelism": f"clip-per-gpu x{world}",
                   "weights": "synthetic (seeded counter-based generator)", "operands": ("MX-fp8 (e4m3 + E8M0 block-32 scales) MFMA for qkv/o_proj/up/down, fp16 attention/patch/CLS tail, "
                                "fp32 accumulate/residual; head fp32") if args.precision == 2 else
                               ("fp32 end to end (the reference's CPU arithmetic): fp32 weights and activations, every contraction on "
                                "v_mfma_f32_16x16x4_f32, fp32 attention / LayerNorm; head fp32") if args.precision == 3 else
                               ("fp32 storage, attention, LayerNorm and element-wise arithmetic as precision 3; GEMM operands split into "
                                "fp16 hi + lo halves (22 bits) by the producing kernels, products a_hi w_hi + a_hi w_lo + a_lo w_hi on "
                                "v_mfma_f32_16x16x32_f16 with fp32 accumulation; head fp32") if args.precision == 4 else
                               "fp16 MFMA, fp32 accumulate/residual; head fp32",
                   "encoder_gflop_per_frame": round(cfg.flops_per_frame(args.hw, args.hw) / 1e9, 3),
                   "head_gflop_per_frame": round(hcfg.flops_per_frame_naive() / 1e9, 4)},
        "end_to_end_tflops": round(value * flops_frame / 1e12, 2),
    }
    if dt_host is not None:
        out["hbm_resident"] = {
            "value": round(hbm_value, 2), "unit": "frames/s", "ms_per_step": round(dt / K * 1e3, 4),
            "what": "the same K steps with the uint8 frames already in HBM when the clock starts and the CLS rows / "
                    "probabilities left in HBM (gathered to rank 0's HBM with N > 1); no PCIe traffic in the timed region",
            "h2d_bytes_per_frame_of_value": args.hw * args.hw * (1 if args.host_input == "green" else 3),
            "host_input_of_value": args.host_input, "bit_identical_to_value_pass": host_equal}
    if dt_host is not None and dt_r3 is not None:
        out["value_r3_definition"] = {"value": round(frames_total / dt_r3, 2), "unit": "frames/s", "ms_per_step": round(dt_r3 / K * 1e3, 4),
                                      "what": "`value` as rounds 1-3 defined it: --host-input rgb (150 528 B per frame over PCIe, green picked on "
                                              "the device), W warm-up steps and NO untimed pre-roll before the K timed steps"}
    if files is not None:
        out["files_path"] = files
    if not args.no_gates and not hung:
        out["gates"] = gates(enc, head, args.model, args.hw, args.precision)
        # `value` is the 16-bit configuration BASELINE.json names; whether ITS labels are the reference's is stated here, next
        # to it, so that nobody reads the fp16 figure as meeting the label half of the contract (label_exact does)
        out["labels_identical"] = labels_identical(out["gates"])
    # Secondary leg, N = 1 only: the SAME model, head and clip through precision 4 - fp32 storage / attention / LayerNorm with
    # the GEMM products as three-term fp16 splits - the mode whose argmax labels are the reference's.  `value`
    # above stays the default fp16-operand mode's, as BASELINE.json's config asks; this says what label identity costs.
    if (world == 1 and args.precision == 0 and not args.no_label_exact and not args.no_gates and not hung
            and (args.model, args.hw) == ("vitb16", 224)):
        enc4 = None
        try:
            enc4 = DinoEncoder.from_weights(cfg, W.synth_encoder_weights(cfg, 1234), device, max_batch=B,
                                            max_frame=(args.hw, args.hw), precision=4)
            enc4.set_lanes(args.lanes)
            k4 = min(K, 40)
            st4 = ClipStream(enc4, head, capacity=k4 * B, classify_every=1024)

            def run4():
                st4.reset()
                for s_ in range(k4):
                    o = (s_ * B) % n_res
                    st4.push_u8(clip[o:o + B])
                return st4.finish()
            run4()
            torch.cuda.synchronize(device)
            t0 = time.perf_counter()
            run4()
            torch.cuda.synchronize(device)
            dt4 = time.perf_counter() - t0
            st4.close()
            out["label_exact"] = {"precision": 4, "value": round(k4 * B / dt4, 2), "unit": "frames/s", "steps": k4,
                                  "ms_per_step": round(dt4 / k4 * 1e3, 4),
                                  "what": "the same model, head and clip (frames resident in HBM) in precision 4: fp32 storage, attention, "
                                          "LayerNorm; GEMM and attention products on the fp16 matrix pipe from operands split into "
                                          "two fp16 halves; one untimed pass, then k steps timed",
                                  "gates": gates(enc4, head, args.model, args.hw, 4)}
            out["label_exact"]["labels_identical"] = labels_identical(out["label_exact"]["gates"])
        except Exception as e:  # noqa: BLE001 - the headline line must not depend on the secondary leg
            out["label_exact"] = {"precision": 4, "error": f"{type(e).__name__}: {e}"}
        finally:
            if enc4 is not None:
                enc4.close()
    if prof:
        gemm = [k for k in prof if k.endswith("_gemm")]
        g_ms = sum(prof[k]["ms"] for k in gemm)
        g_fl = sum(prof[k]["flops"] for k in gemm)
        g_n = sum(prof[k]["launches"] for k in gemm)
        achieved = g_fl / (g_ms * 1e-3) / 1e12
        if args.precision == 4:
            achieved *= 3.0        # the matrix pipe executes three fp16 MFMAs per algorithmic product: this is ITS work
        peak = {2: MFMA_FP8_PEAK_TFLOPS, 3: MFMA_F32_PEAK_TFLOPS}.get(args.precision, MFMA_F16_PEAK_TFLOPS)
        out["roofline"] = {
            "bound": "mfma", "kernel": "gemm_f32_vit_kernel (v_mfma_f32_16x16x4_f32; all epilogues: patch/qkv/o_proj/up/down)"
                                       if args.precision == 3 else
                                       "gemm_split_pp_kernel<EPI, ...> (the ping-pong kernel's split-operand form, gemm_f16_8ph.hip: three v_mfma_f32_16x16x32_f16 per product; "
                                       "`achieved` counts the EXECUTED MFMA work = 3 x the algorithmic FLOPs)" if args.precision == 4 else
                                       "gemm_f16_8ph_kernel (all epilogues: patch/qkv/o_proj/up/down)" +
                                       (", MX-fp8 form (F8 = true)" if args.precision == 2 else ""),
            "achieved": round(achieved, 2), "peak": peak, "unit": "TFLOP/s",
            "frac": round(achieved / peak, 4), "traffic": None, "traffic_source": None,
            "avg_launch_us": round(g_ms * 1e3 / g_n, 2), "launches": g_n,
            "measured": "HIP events around every launch, second pass over the same K steps, one batch in flight",
            "ms_per_step_with_events": round(dt_events / K * 1e3, 4),
            "share_of_timed_region": round(g_ms * 1e-3 / dt_events, 4),
            "by_kernel": {k: {"avg_us": round(v["ms"] * 1e3 / v["launches"], 2),
                              "tflops": round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 1) if v["flops"] else None,
                              "share": round(v["ms"] * 1e-3 / dt_events, 4)} for k, v in prof.items()},
        }
        # the committed PMC passes cover the headline workload in its default and fp32 modes; other workloads report null
        pmc_file, pmc_key, pmc_cmd = {0: ("pmc_traffic.json", "gemm_f16_hbm_bytes_per_launch",
                                          "bench.py itself, one batch in flight, 2 + 20 steps, frames resident "
                                          "(scripts/profile_pmc_bench.sh)"),
                                      3: ("r04_pmc_traffic_fp32.json", "gemm_f32_hbm_bytes_per_launch",
                                          "scripts/quick_perf.py vitb16 64 3 224 3"),
                                      4: ("r05_pmc_traffic_p4.json", "gemm_split_hbm_bytes_per_launch",
                                          "bench.py --precision 4 itself, one batch in flight, 2 + 20 steps, frames resident "
                                          "(PRECISION=4 scripts/profile_pmc_bench.sh)")}.get(args.precision, (None, None, None))
        pmc = os.path.join(HERE, "profiles", pmc_file) if pmc_file else None
        if pmc and os.path.exists(pmc) and (args.model, args.hw, B) == ("vitb16", 224, 64):
            try:
                out["roofline"]["traffic"] = json.load(open(pmc)).get(pmc_key)
                out["roofline"]["traffic_source"] = (f"profiles/{pmc_file}: separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over "
                                                     f"{pmc_cmd} - the same encoder, batch and kernels as this command, frames "
                                                     "resident; taken in its own runs, not in this one")
            except Exception:  # noqa: BLE001
                pass
    if world == 1 and not args.no_cpu_baseline and not hung:
        out["cpu_baseline"] = cpu_baseline(args.model, args.hw, args.cpu_frames, 8)
        if (args.model, args.hw) == ("vitb16", 224):
            # BASELINE.md section 4 / configs[0]: the reference's own CPU-runnable case (ViT-S/16, 64 frames, batch 8) beside it
            out["cpu_baseline"]["cfg1_vits16"] = cpu_baseline("vits16", 224, 64, 8)
    if hung:
        out["error"] = "files_path pass hung (watchdog fired): gates and cpu_baseline skipped, exit code 3"
    finished.set()
    emit(json.dumps(out))
    if hung:
        os._exit(3)


if __name__ == "__main__":
    main()
