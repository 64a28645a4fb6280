"""Build the native library in-tree with hipcc for gfx950 (cross-compiles without a GPU).

    python -m cbas_amd.build [--force] [--verbose] [--debug | --all] [--no-asm-check]

Two variants of the same sources:

  libcbas_mi355x.so         the PRODUCT: exactly the entry points of include/cbas_mi355x.h (what a CBAS installation,
                            bench.py and __graft_entry__.smoke() load)
  libcbas_mi355x_debug.so   the product + the bring-up / test / harness entry points of include/cbas_mi355x_debug.h
                            (-DCBAS_BUILD_DEBUG=1; what the GPU test suite and scripts/ load: CBAS_BUILD_DEBUG=1 in the
                            environment selects it, see _lib.py)

Sources that never mention CBAS_BUILD_DEBUG compile to the same object for both and are compiled once.  After the link,
`asmcheck` disassembles the device code of the kernels that run beside other work and reports the instruction patterns
DESIGN.md section 4 says to keep out of them; its report is written next to the library (asmcheck_report.json).
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
DEBUG_LIB_NAME = "libcbas_mi355x_debug.so"
DEBUG_LIB_PATH = os.path.join(HERE, DEBUG_LIB_NAME)
SOURCES = ["gemm_f16.hip", "gemm_f16_8ph.hip", "gemm_f16_skinny.hip", "gemm_f32.hip", "vit_f32.hip", "vit_kernels.hip", "head_kernels.hip", "head_train_kernels.hip",
           "api_enc.hip", "api_head.hip", "api_head_train.hip", "api_fused.hip", "host_text.cpp", "host_mjpeg.cpp", "host_pixels.cpp"]
DEBUG_ONLY_SOURCES = ["api_debug.hip"]            # harnesses: never in the product
# -packed-fp32-ops: no v_pk_*_f32 at all in the head's kernels (VALU-light; nothing to gain from packed math) - the blunt way to
# keep the instruction form asmcheck bans out of kernels that run beside the encoder and beside training (the host pass
# of the same compile prints "not a recognized feature for this target": expected, harmless)
NO_PACKED = ["-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops"]
EXTRA_FLAGS = {"host_mjpeg.cpp": ["-mavx2"],      # host-only file: 8-lane integer vectors in the inverse DCT (checked at run time)
               "head_kernels.hip": NO_PACKED, "head_train_kernels.hip": NO_PACKED}
ARCH = "gfx950"
HEADERS = [os.path.join(HERE, "..", "include", "cbas_mi355x.h"), os.path.join(HERE, "..", "include", "cbas_mi355x_debug.h")]


def debug_selected() -> bool:
    """CBAS_BUILD_DEBUG=1 in the environment: load / build the debug variant."""
    return os.environ.get("CBAS_BUILD_DEBUG", "0") not in ("", "0")


def lib_path(debug: bool | None = None) -> str:
    return DEBUG_LIB_PATH if (debug_selected() if debug is None else debug) else LIB_PATH


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC or install ROCm)")


def _deps() -> list:
    return [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + HEADERS


def _stale(debug: bool | None = None) -> bool:
    path = lib_path(debug)
    if not os.path.exists(path):
        return True
    t = os.path.getmtime(path)
    return any(os.path.getmtime(d) > t for d in _deps())


def _mentions_debug(src: str) -> bool:
    with open(os.path.join(CSRC, src), "r", errors="replace") as f:
        return "CBAS_BUILD_DEBUG" in f.read()


def build_library(force: bool = False, verbose: bool = False, debug: bool | None = None, asm_check: bool = True) -> str:
    debug = debug_selected() if debug is None else debug
    out_path = lib_path(debug)
    if not force and not _stale(debug):
        return out_path
    hipcc = _hipcc()
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    common = [f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC",
              "-Wno-unused-result", "-I", CSRC]
    headers = [os.path.join(CSRC, hname) for hname in os.listdir(CSRC) if hname.endswith(".h")] + HEADERS
    objs = []
    procs = []
    for src in SOURCES + (DEBUG_ONLY_SOURCES if debug else []):
        variant = debug and (src in DEBUG_ONLY_SOURCES or _mentions_debug(src))
        obj = os.path.join(objdir, os.path.splitext(src)[0] + (".debug.o" if variant else ".o"))
        objs.append(obj)
        srcp = os.path.join(CSRC, src)
        if (not force and os.path.exists(obj) and os.path.getmtime(obj) > os.path.getmtime(srcp)
                and all(os.path.getmtime(obj) > os.path.getmtime(hp) for hp in headers)):
            continue
        cmd = [hipcc, *common, *EXTRA_FLAGS.get(src, []), *(["-DCBAS_BUILD_DEBUG=1"] if variant else []), "-c", srcp, "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    for src, p in procs:
        outp, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{outp}")
        if verbose and outp.strip():
            print(outp)
    cmd = [hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", f"-Wl,--version-script={os.path.join(CSRC, 'exports.map')}",
           "-o", out_path, *objs]
    if verbose:
        print(" ".join(cmd), flush=True)
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stdout}")
    if asm_check and not debug:
        from . import asmcheck
        asmcheck.check_library(out_path, verbose=verbose)      # raises when a kernel contains the banned packed form; writes the report
    elif asm_check:
        # the debug variant is what the GPU suite runs beside MFMA neighbours: its debug-only objects (harness kernels) are held
        # to the same rule; the objects it shares with the product were checked by the product's build
        from . import asmcheck
        rep = asmcheck.check_objects([o for o in objs if o.endswith(".debug.o")])
        if rep["R1"]:
            raise RuntimeError("asmcheck: debug-only kernels contain the banned packed form: " +
                               "; ".join(f"{x['object']}: {x['kernel']}: {x['text']}" for x in rep["R1"][:8]))
    return out_path


def probe_library(path: str, names) -> dict:
    """dlopen ``path`` in a CHILD process and report {"abi", "debug_build", "missing"}.  Not in this process: a library mapped
    BEFORE torch binds to the system's libamdhip64, torch then brings its own, and with two HIP runtimes in one process the
    first HIP call fails with "no ROCm-capable device is detected" (cbas_amd/_lib.py imports torch first for that reason).
    Round 5's build() dlopen'ed both variants ahead of torch and a smoke() in the same process then failed on the GPU box;
    measured afterwards: with torch imported first, even both variants in one process work."""
    import json
    code = ("import ctypes, json, sys\n"
            "lib = ctypes.CDLL(sys.argv[1])\n"
            "names = json.loads(sys.stdin.read())\n"
            "out = {'abi': int(lib.cbas_abi_version()), 'missing': [n for n in names if not hasattr(lib, n)],\n"
            "       'debug_build': int(lib.cbas_debug_build()) if hasattr(lib, 'cbas_debug_build') else 0}\n"
            "print(json.dumps(out))\n")
    r = subprocess.run([sys.executable, "-c", code, path], input=json.dumps(list(names)), capture_output=True, text=True, timeout=120)
    if r.returncode != 0:
        raise RuntimeError(f"could not load {path}:\n{r.stderr}")
    return json.loads(r.stdout.strip().splitlines()[-1])


def build_all(force: bool = False, verbose: bool = False) -> list:
    return [build_library(force, verbose, debug=False), build_library(force, verbose, debug=True)]


if __name__ == "__main__":
    force = "--force" in sys.argv
    verbose = "--verbose" in sys.argv or "-v" in sys.argv
    check = "--no-asm-check" not in sys.argv
    if "--all" in sys.argv:
        for path in [build_library(force, verbose, debug=False, asm_check=check), build_library(force, verbose, debug=True)]:
            print(path)
    else:
        print(build_library(force, verbose, debug=True if "--debug" in sys.argv else None, asm_check=check))
