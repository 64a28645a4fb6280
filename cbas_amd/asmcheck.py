"""Build-time check of the gfx950 device code: the packed-fp32 instruction form that round 5 found computing wrong values.

    python -m cbas_amd.asmcheck [--verbose] [--no-enforce]     (also run by `python -m cbas_amd.build` after the product link)

What it guards against ("The co-residency corruption, root-caused"; scripts/expand_rootcause.py;
profiles/r05_expand_rootcause.json).  Round 4's head_expand_kernel returned wrong values in lanes 48-63 of one wave when waves
of another kernel kept the matrix pipe busy with 32x32x16 MFMAs on the same CU.  Round 5 ran nine single-edit variants of
that kernel's ASSEMBLY in place of the library's kernel, ~78 000 launches each beside the same neighbour, with every wrong
row captured and inverted on the host.  Every wrong value was  (a - b) - b  where  (a - b) - (b - c)  was due: the LOW half of

    v_pk_add_f32 v[8:9], v[10:11], v[8:9] op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]      ; {b - c, a - b}

came out as b - 0 in the last 16 lanes (the instruction's fourth pass), the high half right.  It failed with 8 wait states
before it (operands long since landed), with 8 wait states after it (consumer far away), with the result in a fresh register
pair, with the operands taken from registers instead of LDS, with branch-free code around it - and never (0 of 651 000
launches against 39 of 653 000) when the same subtraction was done by scalar v_sub_f32 or by a v_pk_add_f32 WITHOUT
cross-half operand selection (operands moved into place first).  So the rule is about one instruction form, not about timing,
EXEC masks or LDS:

  R1  (build fails)  a v_pk_{add,mul,fma}_f32 whose op_sel has a 1: its LOW result half is formed from the HIGH half of a
      source register pair.  No kernel of the library may contain one.  (The stand-alone reproducer narrows the fault to
      the selection on SRC1 - add, multiply and FMA: 2, 16 and 9 events - with src0's and the FMA addend's selection clean
      at 1.5e12 / 1.4e12 executions; the rule bans the selection on any source: nothing in the library needs it.)  v_pk_mov_b32 with op_sel is banned with them although
      the stand-alone reproducer (scripts/probes/probe_pk_crosshalf.hip) clears it - 0 wrong results in 2.4e12 executions where
      the arithmetic form had 18 events - because the ban costs three scalar adds in one kernel and "the same operand selection"
      is too close to argue about.
  R2  (reported)     the mirror form, op_sel_hi with a 0 on a VGPR pair (HIGH result half from a LOW source half: the
      scalar-broadcast form the compiler uses everywhere).  Not observed to fail - the GEMM epilogues are full of it and
      every bit-exactness test and soak of rounds 1-4 ran through them, and the stand-alone reproducer ran the form on an add, a
      multiply and an FMA 1.2e12 executions each beside the MFMA loop without a wrong result (where the op_sel forms had 2 and 9 events) -
      counted so that a change in its use is visible.

The report is written next to the library (asmcheck_report.json, git-ignored; tests/test_host_logic.py runs the check).
"""
from __future__ import annotations

import json
import os
import re
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
LLVM_BIN = os.environ.get("CBAS_LLVM_BIN", "/opt/rocm/lib/llvm/bin")
TARGET = "hipv4-amdgcn-amd-amdhsa--gfx950"
REPORT = os.path.join(HERE, "asmcheck_report.json")

_INSN = re.compile(r"^\s+([a-z_0-9]+)\s*(.*?)\s*//\s*([0-9A-Fa-f]+):")
_FUNC = re.compile(r"^[0-9a-f]+ <([^>]+)>:$")
_PK_F32 = re.compile(r"^v_pk_((add|mul|fma)_f32|mov_b32)$")      # v_pk_mov_b32 shares the operand selection (see R1)
_OP_SEL = re.compile(r"\bop_sel:\[([01,]+)\]")
_OP_SEL_HI = re.compile(r"\bop_sel_hi:\[([01,]+)\]")


def disassemble_object(obj_path: str) -> dict:
    """{kernel name: [(mnemonic, operand text, address)]} of the gfx950 code object bundled in one .o (or a .so with one TU)."""
    with tempfile.TemporaryDirectory() as td:
        fb, co = os.path.join(td, "fb.bin"), os.path.join(td, "dev.co")
        r = subprocess.run([os.path.join(LLVM_BIN, "llvm-objcopy"), "-O", "binary", "--only-section=.hip_fatbin", obj_path, fb],
                           capture_output=True, text=True)
        if r.returncode != 0 or not os.path.exists(fb) or os.path.getsize(fb) == 0:
            return {}                                          # a host-only object
        r = subprocess.run([os.path.join(LLVM_BIN, "clang-offload-bundler"), "--unbundle", "--type=o", f"--input={fb}",
                            f"--targets={TARGET}", f"--output={co}"], capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"clang-offload-bundler failed on {obj_path}: {r.stderr}")
        r = subprocess.run([os.path.join(LLVM_BIN, "llvm-objdump"), "-d", co], capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"llvm-objdump failed on {obj_path}: {r.stderr}")
    return parse_disassembly(r.stdout)


def parse_disassembly(text: str) -> dict:
    kernels, cur = {}, None
    for line in text.splitlines():
        m = _FUNC.match(line)
        if m:
            cur = kernels.setdefault(m.group(1), [])
            continue
        if cur is None:
            continue
        m = _INSN.match(line)
        if m:
            cur.append((m.group(1), m.group(2), m.group(3)))
    return {k: v for k, v in kernels.items() if v}


def check_kernel(insns: list) -> list:
    """Findings of one kernel: {rule, addr, text}."""
    out = []
    for mn, ops, addr in insns:
        if not _PK_F32.match(mn):
            continue
        m = _OP_SEL.search(ops)
        if m and "1" in m.group(1):
            out.append({"rule": "R1", "addr": addr, "text": f"{mn} {ops}"})
            continue
        m = _OP_SEL_HI.search(ops)
        if m:
            srcs = [s.strip() for s in ops.split(",")[1:]]     # sources in order; only a VGPR PAIR has a low half to re-use
            bits = m.group(1).split(",")
            if any(b == "0" and i < len(srcs) and srcs[i].lstrip("-|").startswith("v[") for i, b in enumerate(bits)):
                out.append({"rule": "R2", "addr": addr, "text": f"{mn} {ops}"})
    return out


def check_objects(obj_paths: list) -> dict:
    report = {"kernels": 0, "instructions": 0, "packed_f32_ops": 0, "R1": [], "R2_count": 0, "R2_kernels": 0}
    for obj in obj_paths:
        for name, insns in disassemble_object(obj).items():
            report["kernels"] += 1
            report["instructions"] += len(insns)
            report["packed_f32_ops"] += sum(1 for mn, _, _ in insns if _PK_F32.match(mn))
            fs = check_kernel(insns)
            report["R1"] += [dict(f, kernel=name, object=os.path.basename(obj)) for f in fs if f["rule"] == "R1"]
            n2 = sum(1 for f in fs if f["rule"] == "R2")
            report["R2_count"] += n2
            report["R2_kernels"] += int(n2 > 0)
    return report


def product_objects() -> list:
    objdir = os.path.join(HERE, "build")
    return sorted(os.path.join(objdir, f) for f in os.listdir(objdir) if f.endswith(".o") and not f.endswith(".debug.o"))


def check_library(lib_path: str | None = None, verbose: bool = False, enforce: bool = True) -> dict:
    """Check every object the product library was linked from (its fat binary is the concatenation of their bundles).
    Writes asmcheck_report.json; raises when any kernel contains an R1 instruction."""
    report = check_objects(product_objects())
    report["library"] = os.path.basename(lib_path) if lib_path else None
    with open(REPORT, "w") as f:
        json.dump(report, f, indent=1)
    if verbose:
        print(json.dumps({k: v for k, v in report.items() if k != "R1"}, indent=1))
    if enforce and report["R1"]:
        lines = "\n".join(f"  {x['object']}: {x['kernel']}: {x['text']}" for x in report["R1"][:24])
        raise RuntimeError(f"asmcheck: {len(report['R1'])} packed-fp32 instruction(s) form their low half from the high half of a "
                           f"source pair (op_sel) - the form that computes wrong values beside MFMA-heavy neighbours "
                           f"(DESIGN.md section 4):\n{lines}")
    return report


if __name__ == "__main__":
    rep = check_library(verbose="--verbose" in sys.argv or "-v" in sys.argv, enforce="--no-enforce" not in sys.argv)
    print(json.dumps({"kernels": rep["kernels"], "instructions": rep["instructions"], "packed_f32_ops": rep["packed_f32_ops"],
                      "R1": len(rep["R1"]), "R2_count": rep["R2_count"], "R2_kernels": rep["R2_kernels"]}))
