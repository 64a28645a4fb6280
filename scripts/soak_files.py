"""Soak of the file path: many clips of mixed lengths and formats, back to back through cbas_amd.dist.encode_files (one process),
twice; every `_cls.h5` / `_outputs.csv` must be byte-identical between the passes and to the same clip run alone through
encode_infer_file.  Exercises the decode-ahead threads, the page-locked rings, the pipelined clips and the writer threads.

    python scripts/soak_files.py [repeats] [out.json]          (CBAS_SOAK_PRECISION=4: the same through precision 4)
"""
import hashlib
import json
import os
import shutil
import sys
import tempfile
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cbas_amd import config as C, dist as cdist, framesource as F, pipeline as P, synth, weights as W  # noqa: E402
from cbas_amd.encoder import DinoEncoder  # noqa: E402
from cbas_amd.head import ClassifierLSTMDeltas  # noqa: E402


def sha(path):
    return hashlib.sha256(open(path, "rb").read()).hexdigest()


def main():
    repeats = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    cfg = C.VIT_B16
    precision = int(os.environ.get("CBAS_SOAK_PRECISION", "0"))          # 4: the label-exact mode's kernels under the same load
    enc = DinoEncoder.from_weights(cfg, W.synth_encoder_weights(cfg, 1234), "cuda", max_batch=64, max_frame=(224, 224), precision=precision)
    head = ClassifierLSTMDeltas(768, 9)
    head.load_state_dict(W.synth_head_weights(C.HeadConfig(), 4321))
    head.to("cuda")
    names = [f"b{i}" for i in range(9)]
    root = tempfile.mkdtemp(prefix="cbas_soak_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    lengths = [700, 1, 4096, 31, 2049, 513, 128, 3000]
    base = synth.cage_frames(3, 600, 224, 224)
    protos = []
    for k, n in enumerate(lengths):
        fr = base[(np.arange(n) * (k + 1)) % 600]                       # different content per clip
        p = os.path.join(root, f"proto{k}." + ("avi" if k % 2 == 0 else "npy"))
        if p.endswith(".avi"):
            F.write_mjpeg_avi(p, fr, quality=85, subsampling=2)
        else:
            np.save(p, fr)
        protos.append(p)
    # reference outputs: each clip alone
    alone = []
    for k, p in enumerate(protos):
        d = os.path.join(root, f"alone{k}")
        os.makedirs(d)
        q = os.path.join(d, os.path.basename(p))
        os.symlink(p, q)
        h5, csv = P.encode_infer_file(enc, head, q, "soak", names)
        alone.append((sha(h5), sha(csv)))
    paths = []
    for r in range(repeats):
        order = np.random.default_rng(r).permutation(len(protos))
        for k in order:
            d = os.path.join(root, f"r{r}_{k}")
            os.makedirs(d)
            q = os.path.join(d, os.path.basename(protos[k]))
            os.symlink(protos[k], q)
            paths.append((int(k), q))
    res = {"precision": precision, "clips": len(paths), "frames": int(sum(lengths[k] for k, _ in paths)), "passes": []}
    ok = True
    n_pass = int(os.environ.get("CBAS_SOAK_PASSES", "2"))

    def rss_mb():
        for line in open("/proc/self/status"):
            if line.startswith("VmRSS"):
                return int(line.split()[1]) / 1024.0
        return 0.0
    for ps in range(n_pass):
        t0 = time.perf_counter()
        recs = cdist.encode_files([q for _, q in paths], enc, head=head, dataset_name="soak", behaviors=names)
        dt = time.perf_counter() - t0
        bad = 0
        for (k, _q), r in zip(paths, recs):
            if r["status"] != "ok" or (sha(r["cls_file"]), sha(r["csv_file"])) != alone[k]:
                bad += 1
        ok &= bad == 0
        for r in recs:                                   # the next pass writes the same files again
            os.remove(r["cls_file"])
            os.remove(r["csv_file"])
        free, total = torch.cuda.mem_get_info()
        res["passes"].append({"seconds": round(dt, 3), "frames_per_s": round(res["frames"] / dt), "clips_differing_from_alone": bad,
                              "host_rss_mb": round(rss_mb()), "hbm_used_mb": round((total - free) / 2 ** 20)})
        print(res["passes"][-1], flush=True)
    res["all_identical"] = bool(ok)
    head.close()
    enc.close()
    shutil.rmtree(root, ignore_errors=True)
    print(json.dumps(res))
    if len(sys.argv) > 2:
        os.makedirs(os.path.dirname(os.path.abspath(sys.argv[2])), exist_ok=True)
        json.dump(res, open(sys.argv[2], "w"), indent=1)
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
