"""Build an experiment variant of the library: the GEMM sources recompiled with extra -D flags, everything else
reused from cbas_amd/build/.  Output: cbas_amd/libcbas_exp_<name>.so (load it with CBAS_EXP_LIB=<path> in the
scripts that support it).

    python scripts/build_exp.py <name> -DFLAG [-DFLAG ...]
"""
import os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cbas_amd import build as B

name, flags = sys.argv[1], sys.argv[2:]
B.build_library()
hipcc = B._hipcc()
objdir = os.path.join(B.HERE, "build")
expdir = os.path.join(objdir, "exp_" + name)
os.makedirs(expdir, exist_ok=True)
objs, procs = [], []
for src in B.SOURCES:
    if src.startswith("gemm_f16"):
        obj = os.path.join(expdir, src.replace(".hip", ".o"))
        procs.append(subprocess.Popen([hipcc, f"--offload-arch={B.ARCH}", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-result",
                                       "-I", B.CSRC, *flags, "-c", os.path.join(B.CSRC, src), "-o", obj]))
    else:
        obj = os.path.join(objdir, src.replace(".hip", ".o"))
    objs.append(obj)
assert all(p.wait() == 0 for p in procs)
out = os.path.join(B.HERE, f"libcbas_exp_{name}.so")
subprocess.check_call([hipcc, f"--offload-arch={B.ARCH}", "-shared", "-fPIC", "-o", out, *objs])
print(out)
