"""Decode / copy / compute overlap of the file path, from a rocprofv3 trace.

Step 1 (on the GPU box, under the profiler - the program itself after `--`):
    rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d gpurun_out/tl -o tl -- \
        python scripts/files_timeline.py run [avi|npy] [frames]
  encodes + classifies one clip twice through cbas_amd.pipeline.encode_infer_file (the first pass warms buffers and
  sessions), sleeping 0.4 s before the second so that its activity is the trace's last cluster.
Step 2:
    python scripts/files_timeline.py report gpurun_out/tl [out.json]
  takes the last cluster of GPU activity and reports, for the span from its first copy to its last: the fraction with a kernel
  running, with a host->device copy running, with both, and the share of the copy time that ran under kernels.
"""
import csv
import glob
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run(fmt: str, n: int, clips: int = 1) -> None:
    import tempfile
    import numpy as np
    import torch
    from cbas_amd import config as C, framesource as F, pipeline as P, weights as W
    from cbas_amd.encoder import DinoEncoder
    from cbas_amd.head import ClassifierLSTMDeltas
    dev = torch.device("cuda:0")
    cfg = C.NAMED_VIT["vitb16"]
    enc = DinoEncoder.from_weights(cfg, W.synth_encoder_weights(cfg, 1234), dev)
    hcfg = C.HeadConfig(in_features=cfg.hidden_size, out_features=9)
    head = ClassifierLSTMDeltas(cfg.hidden_size, 9, seq_len=31)
    head.load_state_dict(W.synth_head_weights(hcfg, 4321))
    head.to(dev)
    gen = torch.Generator(device=dev)
    gen.manual_seed(1)
    low = torch.rand((n, 1, 14, 14), device=dev, generator=gen)
    g = torch.nn.functional.interpolate(low, size=(224, 224), mode="bicubic", align_corners=False) * 200.0 + 28.0
    g8 = (g + torch.randn((n, 1, 224, 224), device=dev, generator=gen) * 4.0).clamp_(0, 255).to(torch.uint8)[:, 0]
    fr = g8[..., None].expand(n, 224, 224, 3).contiguous().cpu().numpy()
    root = tempfile.mkdtemp(prefix="cbas_tl_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    path = os.path.join(root, "clip." + fmt)
    if fmt == "avi":
        F.write_mjpeg_avi(path, fr, quality=85, subsampling=2)
    else:
        np.save(path, fr)
    del fr, g, g8, low
    names = [f"b{i}" for i in range(9)]
    if clips > 1:                                    # the multi-clip product path: N copies of the clip through dist.encode_files
        from cbas_amd import dist as cdist
        paths = []
        for k in range(clips):
            d = os.path.join(root, f"c{k}")
            os.makedirs(d)
            q = os.path.join(d, "clip." + fmt)
            os.symlink(path, q)
            paths.append(q)
        for k in range(2):
            torch.cuda.synchronize()
            time.sleep(0.4)
            t0 = time.perf_counter()
            recs = cdist.encode_files(paths, enc, head=head, dataset_name="tl", behaviors=names)
            dt = time.perf_counter() - t0
            assert all(r["status"] == "ok" for r in recs)
            print(f"pass {k}: {clips} clips, {clips * n / dt:.0f} frames/s ({dt * 1e3:.1f} ms)", flush=True)
        head.close()
        enc.close()
        import shutil
        shutil.rmtree(root, ignore_errors=True)
        return
    for k in range(2):
        torch.cuda.synchronize()
        time.sleep(0.4)
        t0 = time.perf_counter()
        out = P.encode_infer_file(enc, head, path, "tl", names)
        dt = time.perf_counter() - t0
        print(f"pass {k}: {n / dt:.0f} frames/s ({dt * 1e3:.1f} ms) -> {out}", flush=True)
    head.close()
    enc.close()
    import shutil
    shutil.rmtree(root, ignore_errors=True)


def _union(iv):
    iv = sorted(iv)
    out = []
    for s, e in iv:
        if out and s <= out[-1][1]:
            out[-1][1] = max(out[-1][1], e)
        else:
            out.append([s, e])
    return out


def _length(iv):
    return sum(e - s for s, e in iv)


def _intersect(a, b):
    out, i, j = [], 0, 0
    while i < len(a) and j < len(b):
        s, e = max(a[i][0], b[j][0]), min(a[i][1], b[j][1])
        if s < e:
            out.append([s, e])
        if a[i][1] < b[j][1]:
            i += 1
        else:
            j += 1
    return out


def report(d: str, out_json=None) -> None:
    kfile = sorted(glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True))[-1]
    cfile = sorted(glob.glob(os.path.join(d, "**", "*memory_copy_trace.csv"), recursive=True))[-1]
    kern = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(kfile))]
    cop = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Direction", "")) for r in csv.DictReader(open(cfile))]
    ev = sorted([(s, e) for s, e, _ in kern] + [(s, e) for s, e, _ in cop])
    # the last cluster: everything after the last gap of more than 0.2 s
    start = ev[0][0]
    last_end = ev[0][1]
    for s, e in ev:
        if s - last_end > 200_000_000:
            start = s
        last_end = max(last_end, e)
    kern = [(s, e, n) for s, e, n in kern if s >= start]
    cop = [(s, e, n) for s, e, n in cop if s >= start]
    t0 = min(min(s for s, _, _ in kern), min(s for s, _, _ in cop))
    t1 = max(max(e for _, e, _ in kern), max(e for _, e, _ in cop))
    span = t1 - t0
    ku = _union([(s, e) for s, e, _ in kern])
    h2d = _union([(s, e) for s, e, dr in cop if "HOST_TO_DEVICE" in dr.upper()])
    d2h = _union([(s, e) for s, e, dr in cop if "DEVICE_TO_HOST" in dr.upper()])
    both = _intersect(ku, h2d)
    gemm = _union([(s, e) for s, e, n in kern if "gemm_f16_8ph" in n])
    gaps = sorted(((ku[i + 1][0] - ku[i][1]) / 1e6 for i in range(len(ku) - 1)), reverse=True)
    res = {
        "largest_kernel_idle_gaps_ms": [round(g, 3) for g in gaps[:8]], "idle_total_ms": round(sum(gaps), 3),
        "span_ms": span / 1e6, "kernels": len(kern), "copies": len(cop),
        "kernel_busy": _length(ku) / span, "gemm_busy": _length(gemm) / span,
        "h2d_busy": _length(h2d) / span, "d2h_busy": _length(d2h) / span,
        "h2d_under_kernels": _length(both) / max(1, _length(h2d)),
        "first_kernel_after_first_copy_ms": (min(s for s, _, _ in kern) - t0) / 1e6,
        "tail_after_last_h2d_ms": (t1 - max(e for _, e in h2d)) / 1e6 if h2d else None,
        "h2d_bytes_note": "green planes (50 176 B per frame) for avi clips, RGB frames (150 528 B) for npy clips",
    }
    print(json.dumps(res, indent=1))
    if out_json:
        json.dump(res, open(out_json, "w"), indent=1)


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run(sys.argv[2] if len(sys.argv) > 2 else "avi", int(sys.argv[3]) if len(sys.argv) > 3 else 8192,
            int(sys.argv[4]) if len(sys.argv) > 4 else 1)
    else:
        report(sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else None)
