"""Encode (and optionally classify) a list of videos on every GPU of the node, one process per GPU:

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 -m cbas_amd.encode_files \
        --encoder <checkpoint dir | HF id in the local cache> [--model-bundle <dir with model.pth + model_meta.json>] \
        [--dataset-name NAME] video1.mp4 video2.mp4 ...        (or --dir <recordings root>)

What the reference does with its EncodeThread queue and, when a model is live, its ClassificationThread queue
(backend/workthreads.py:276-348, 453-519) on one device: here every rank pulls the next video from one shared queue,
ships its rows to rank 0 over RCCL, and rank 0's writer threads produce ``<video>_cls.h5`` /
``<video>_<dataset>_outputs.csv`` exactly as ``encode_file`` / ``infer_file`` do (cbas_amd/dist.py).  Works unchanged
with one process (no torchrun).  With ``--dir`` the videos are those the reference would queue on project load
(startup_page.py:80-126): every ``*.mp4`` without an up-to-date ``_cls.h5``.
"""
from __future__ import annotations

import argparse
import os
import sys
from typing import List

import torch


def _needs_encoding(video: str, stamp: str) -> bool:
    """startup_page.py:92-117: missing, unreadable, unstamped or differently stamped `_cls.h5` -> (re)encode."""
    from . import h5io
    h5 = os.path.splitext(video)[0] + "_cls.h5"
    if not os.path.exists(h5):
        return True
    try:
        with h5io.ClsReader(h5) as r:
            return r.attrs.get("encoder_model_identifier") != stamp
    except Exception:  # noqa: BLE001
        return True


def find_videos(root: str, stamp: str) -> List[str]:
    out = []
    for d, _sub, files in os.walk(root):
        for f in sorted(files):
            if f.lower().endswith(".mp4") and _needs_encoding(os.path.join(d, f), stamp):
                out.append(os.path.join(d, f))
    return sorted(out)


def main(argv=None) -> int:
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("videos", nargs="*")
    ap.add_argument("--dir", default=None, help="recordings root: encode every .mp4 lacking an up-to-date _cls.h5")
    ap.add_argument("--encoder", required=True, help="encoder_model_identifier: checkpoint directory or cached HF id")
    ap.add_argument("--model-bundle", default=None, help="directory with model.pth + config.yaml + model_meta.json")
    ap.add_argument("--dataset-name", default=None, help="<video>_<dataset-name>_outputs.csv (default: the bundle's name)")
    ap.add_argument("--split-clips", choices=("auto", "always", "never"), default="auto",
                    help="split EACH video's frames over all GPUs (halo exchange at the cuts) instead of giving each GPU "
                         "whole videos; auto: when there are fewer videos than GPUs")
    ap.add_argument("--local-writes", action="store_true",
                    help="every rank writes the files of its own videos (one node, one filesystem) instead of sending the rows "
                         "to rank 0: no write queue on rank 0 when there is about one video per GPU")
    ap.add_argument("--max-batch", type=int, default=128)
    ap.add_argument("--max-frame", type=int, nargs=2, default=(256, 256))
    ap.add_argument("--precision", type=int, default=None, choices=(0, 1, 2, 3, 4),
                    help="default: CBAS_PRECISION from the environment, else 4 = fp32 storage / attention / LayerNorm with the GEMM "
                         "products as three-term fp16 splits: rows ~1e-6 from the reference's fp32 CPU path and every argmax "
                         "label the reference's.  0 = the explicit fast mode: fp16 operands (the reference's own GPU behaviour "
                         "under autocast), 2.2x the frame rate, rows within the 1e-3 contract, <= 1 %% of near-tie labels differ "
                         "from the CPU path.  1 fp16 hi+lo weights; 2 MX-fp8 throughput mode (needs --experimental-fp8); 3 fp32 "
                         "end to end on the fp32 matrix pipe (the literal restatement, 1/3 of mode 4's rate)")
    ap.add_argument("--experimental-fp8", action="store_true",
                    help="allow --precision 2: rows are ~6e-2 from the fp32 reference, NOT interchangeable with fp16 rows; "
                         "the files are stamped '<encoder>#mx-fp8' + attr encoder_precision so that CBAS and this tool "
                         "treat them as made by a different encoder")
    args = ap.parse_args(argv)
    if args.precision is None:
        from .encoder import DEFAULT_PRECISION
        args.precision = int(os.environ.get("CBAS_PRECISION", str(DEFAULT_PRECISION)))
    if args.precision == 2 and not args.experimental_fp8:
        ap.error("--precision 2 writes MX-fp8 rows that heads trained on fp16 embeddings must not consume; "
                 "pass --experimental-fp8 to write them (stamped as such)")

    from . import dist as cdist, pipeline as P
    from .bundle import load_model_bundle
    from .encoder import DinoEncoder
    rank, world, local = cdist.init_from_env(os.environ.get("CBAS_DIST_BACKEND"))
    local = local % max(1, torch.cuda.device_count())
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    videos = list(args.videos)
    stamp = args.encoder + ("#" + P.FP8_TAG if args.precision == 2 else "")      # what pipeline.file_attrs writes
    if args.dir:
        videos += find_videos(args.dir, stamp)
    if not videos:
        if rank == 0:
            print("nothing to encode")
        return 0
    P.set_project_stamp(args.encoder)                       # what gui_state.proj.encoder_model_identifier is to encode_file
    enc = DinoEncoder(args.encoder, device=device, max_batch=args.max_batch, max_frame=tuple(args.max_frame),
                      precision=args.precision)
    head = meta = None
    if args.model_bundle:
        # an MX-fp8 run only accepts a head whose bundle says it was trained on MX-fp8 rows (same '#mx-fp8' stamp)
        head, meta = load_model_bundle(args.model_bundle, device=device, project_encoder=stamp,
                                       in_features=enc.config.hidden_size)
        if head is None:
            return 2
    hp = (meta or {}).get("hyperparameters", {})
    name = args.dataset_name or os.path.basename(os.path.normpath(args.model_bundle or "")) or None
    temperature = float((meta or {}).get("calibration", {}).get("temperature", 1.0))        # workthreads.py:484
    split = world > 1 and (args.split_clips == "always" or (args.split_clips == "auto" and len(videos) < world))
    if split:
        # fewer videos than GPUs: every GPU takes a frame range of each video (last sentence)
        ok = 0
        for v in videos:
            try:
                h5, _csv = cdist.encode_infer_file_sharded(v, enc, head=head, dataset_name=name, behaviors=hp.get("behaviors"),
                                                           temperature=temperature)
                ok += int(h5 is not None)
            except Exception as e:  # noqa: BLE001 - EncodeThread logs and goes on (workthreads.py:334-336)
                print(f"ERROR during encoding for {v} on rank {rank}: {e}")
            cdist.barrier()
        if rank == 0:
            print(f"{ok} of {len(videos)} videos encoded, each split over {world} GPU(s)")
    else:
        recs = cdist.encode_files(videos, enc, head=head, dataset_name=name, behaviors=hp.get("behaviors"),
                                  temperature=temperature, local_writes=args.local_writes)
        cdist.barrier()
        if rank == 0:
            ok = sum(r["status"] == "ok" for r in recs)
            print(f"{ok} of {len(recs)} videos encoded on {world} GPU(s); {sum(r['frames'] for r in recs)} frames")
    enc.close()
    return 0


if __name__ == "__main__":
    sys.exit(main())
