"""Build libcbas_mi355x.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

    python -m cbas_amd.build [--force] [--verbose]
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB_NAME = "libcbas_mi355x.so"
LIB_PATH = os.path.join(HERE, LIB_NAME)
SOURCES = ["gemm_f16.hip", "gemm_f16_8ph.hip", "gemm_f16_skinny.hip", "gemm_f32.hip", "vit_f32.hip", "vit_kernels.hip", "head_kernels.hip", "head_train_kernels.hip",
           "api_enc.hip", "api_head.hip", "api_head_train.hip", "api_fused.hip", "host_text.cpp", "host_mjpeg.cpp", "host_pixels.cpp"]
EXTRA_FLAGS = {"host_mjpeg.cpp": ["-mavx2"]}      # host-only file: 8-lane integer vectors in the inverse DCT (checked at run time)
ARCH = "gfx950"


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC or install ROCm)")


def _stale() -> bool:
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(HERE, "..", "include", "cbas_mi355x.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build_library(force: bool = False, verbose: bool = False) -> str:
    if not force and not _stale():
        return LIB_PATH
    hipcc = _hipcc()
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    common = [f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC",
              "-Wno-unused-result", "-I", CSRC]
    objs = []
    procs = []
    for src in SOURCES:
        obj = os.path.join(objdir, os.path.splitext(src)[0] + ".o")
        objs.append(obj)
        srcp = os.path.join(CSRC, src)
        headers = [os.path.join(CSRC, hname) for hname in os.listdir(CSRC) if hname.endswith(".h")]
        headers.append(os.path.join(HERE, "..", "include", "cbas_mi355x.h"))
        if (not force and os.path.exists(obj) and os.path.getmtime(obj) > os.path.getmtime(srcp)
                and all(os.path.getmtime(obj) > os.path.getmtime(hp) for hp in headers)):
            continue
        cmd = [hipcc, *common, *EXTRA_FLAGS.get(src, []), "-c", srcp, "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    for src, p in procs:
        outp, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{outp}")
        if verbose and outp.strip():
            print(outp)
    cmd = [hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", LIB_PATH, *objs]
    if verbose:
        print(" ".join(cmd), flush=True)
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stdout}")
    return LIB_PATH


if __name__ == "__main__":
    path = build_library(force="--force" in sys.argv, verbose="--verbose" in sys.argv or "-v" in sys.argv)
    print(path)
