"""Build-time check of the gfx950 device code: instruction patterns that DESIGN.md section 4 ("An interference the file-path
soak found") says to keep out of kernels that run BESIDE other work.

    python -m cbas_amd.asmcheck [--verbose]          (also run by `python -m cbas_amd.build` after the product link)

Background.  Round 4's head_expand_kernel returned wrong values in lanes 48-63 of one wave when waves of another kernel kept
the matrix pipe busy on the same CU.  In the device code of that kernel the corrupted value - and only it - went through

    s_and_saveexec_b64 ...                      ; EXEC = the lanes of the third stream
    s_waitcnt lgkmcnt(0)
    v_pk_add_f32 v[8:9], v[10:11], v[8:9] op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]
    s_nop 0
    v_sub_f32 v18, v9, v8

i.e. a packed-fp32 result consumed one wait state later, under a partial EXEC mask, with operands fresh from LDS.  Round 5
ran instruction-level variants of that kernel beside the same neighbour (scripts/expand_rootcause.py,
profiles/r05_expand_rootcause.json; DESIGN section 4 states what they showed).  This module is the standing guard: it
disassembles every kernel of the built objects and lists

  R1  a v_pk_*_f32 whose result is read by a VALU instruction <= PK_MIN_WAIT wait states later while EXEC may be partial
  R2  (report only) the same under full EXEC
  R3  (report only) divergent regions (s_and_saveexec ... s_or exec) that contain a transcendental or packed op at all

R1 in a GUARDED kernel (one that shares the device with other streams / processes in CBAS: the head's inference kernels,
the head's training kernels, the encoder's element-wise kernels) fails the build.  The EXEC tracking is a linear scan of the
instruction stream (saveexec / exec-writing SALU ops open a region, `s_or_b64 exec, exec, sN` closes one, `s_mov_b64 exec,
-1` closes all): conservative for the structured code the compiler emits, not a proof.
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
PK_MIN_WAIT = 1                     # wait states between a packed-fp32 producer and its first VALU consumer that R1 / R2 flag

# kernels that run beside other work in CBAS (EncodeThread, ClassificationThread and TrainingThread share one device,
# backend/workthreads.py:1256-1267): substrings of the (mangled) kernel names
GUARDED = ("head_expand", "head_centre", "head_lstm", "head_pool", "f16_to_f32", "train_", "lstm_train", "adam_",
           "layernorm_", "final_norm_cls", "im2col_", "write_prefix", "attention_")

_INSN = re.compile(r"^\s+([a-z_0-9]+)\s*(.*?)\s*//\s*([0-9A-Fa-f]+):")
_FUNC = re.compile(r"^[0-9a-f]+ <([^>]+)>:$")
_LABEL = re.compile(r"^<(L[0-9]+)>:$")
_REG = re.compile(r"\b([vas])\[(\d+):(\d+)\]|\b([vas])(\d+)\b")


def _regs(text: str, kind: str = "v") -> set:
    out = set()
    for m in _REG.finditer(text):
        if m.group(1):
            if m.group(1) == kind:
                out.update(range(int(m.group(2)), int(m.group(3)) + 1))
        elif m.group(4) == kind:
            out.add(int(m.group(5)))
    return out


def _split_operands(ops: str):
    """dst text, src text of an instruction's operand string (modifiers such as op_sel:[..] dropped)."""
    ops = re.sub(r"\b[a-z_0-9]+:\[[^\]]*\]", "", ops)          # op_sel:[0,1] ...
    ops = re.sub(r"\b(offset|clamp|mul|div|row_[a-z]+|quad_perm|bank_mask|bound_ctrl|dst_sel|src[01]_sel|dst_unused|cbsz|abid|blgp|offen|sc[01]|nt)[:\w]*", "", ops)
    parts = [p.strip() for p in ops.split(",")]
    # a register range "v[8:9]" contains no comma, so a plain split is enough
    return (parts[0] if parts else ""), ", ".join(parts[1:])


def disassemble_object(obj_path: str) -> dict:
    """{kernel name: [(mnemonic, operand text, address)] with ("<label>", name, None) entries} of one .o / .so's bundle."""
    with tempfile.TemporaryDirectory() as td:
        fb, co = os.path.join(td, "fb.bin"), os.path.join(td, "dev.co")
        r = subprocess.run([os.path.join(LLVM_BIN, "llvm-objcopy"), "-O", "binary", "--only-section=.hip_fatbin", obj_path, fb],
                           capture_output=True, text=True)
        if r.returncode != 0 or not os.path.exists(fb) or os.path.getsize(fb) == 0:
            return {}
        r = subprocess.run([os.path.join(LLVM_BIN, "clang-offload-bundler"), "--unbundle", "--type=o", f"--input={fb}",
                            f"--targets={TARGET}", f"--output={co}"], capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"clang-offload-bundler failed on {obj_path}: {r.stderr}")
        r = subprocess.run([os.path.join(LLVM_BIN, "llvm-objdump"), "-d", "--symbolize-operands", co], capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"llvm-objdump failed on {obj_path}: {r.stderr}")
    return parse_disassembly(r.stdout)


def parse_disassembly(text: str) -> dict:
    kernels, cur = {}, None
    for line in text.splitlines():
        m = _FUNC.match(line)
        if m and not re.fullmatch(r"L\d+", m.group(1)):
            cur = kernels.setdefault(m.group(1), [])
            continue
        if cur is None:
            continue
        if m:                                               # "<addr> <L12>:" - a branch target (--symbolize-operands)
            cur.append(("<label>", m.group(1), None))
            continue
        m = _LABEL.match(line.strip())
        if m:
            cur.append(("<label>", m.group(1), None))
            continue
        m = _INSN.match(line)
        if m:
            cur.append((m.group(1), m.group(2), m.group(3)))
    return {k: v for k, v in kernels.items() if v}


_TRANS = ("v_exp_", "v_log_", "v_rcp_", "v_rsq_", "v_sqrt_", "v_sin_", "v_cos_")


def _is_valu(mn: str) -> bool:
    return mn.startswith("v_") and not mn.startswith(("v_mfma", "v_smfmac", "v_accvgpr", "v_readlane", "v_readfirstlane"))


def check_kernel(insns: list) -> list:
    """Findings for one kernel: dicts {rule, addr, text, consumer, wait_states, exec_partial}."""
    findings = []
    depth = 0                       # open EXEC regions (linear scan)
    region_has = None               # for R3: first trans / packed op seen inside the current outermost region
    n = len(insns)
    for i, (mn, ops, addr) in enumerate(insns):
        if mn == "<label>":
            continue
        # ---- EXEC tracking -----------------------------------------------------------------------------------
        if "saveexec" in mn:
            depth += 1
        elif mn.startswith("s_") and re.match(r"^exec\b", ops):
            if mn == "s_or_b64":
                depth = max(0, depth - 1)
            elif mn == "s_mov_b64" and re.search(r",\s*-1$", ops):
                depth = 0
            else:                                           # s_and / s_andn2 / s_xor / s_mov exec, sN: lanes switched off
                depth += 1
        elif mn.startswith("v_cmpx"):
            depth += 1
        if depth == 0:
            region_has = None
        partial = depth > 0
        if partial and region_has is None and (mn.startswith("v_pk_") or mn.startswith(_TRANS)):
            region_has = (mn, addr)
            findings.append({"rule": "R3", "addr": addr, "text": f"{mn} {ops}", "consumer": None, "wait_states": None,
                             "exec_partial": True})
        # ---- packed fp32 producer -> consumer distance ---------------------------------------------------------
        if mn.startswith("v_pk_") and mn.endswith("_f32"):
            dst, _ = _split_operands(ops)
            dregs = _regs(dst)
            waits = 0
            for j in range(i + 1, n):
                mj, oj, aj = insns[j]
                if mj == "<label>" or mj.startswith(("s_cbranch", "s_branch", "s_endpgm", "s_setpc", "s_barrier")):
                    break
                if mj == "s_nop":
                    waits += int(oj.strip() or 0) + 1
                    continue
                if waits > PK_MIN_WAIT:
                    break
                if _is_valu(mj):
                    _, src = _split_operands(oj)
                    # v_fmac / v_pk_fma accumulate: the destination is a source too
                    rd = _regs(src) | (_regs(_split_operands(oj)[0]) if mj.startswith(("v_fmac", "v_mac", "v_pk_fmac")) else set())
                    hit = rd & dregs
                    if hit:
                        findings.append({"rule": "R1" if partial else "R2", "addr": addr, "text": f"{mn} {ops}",
                                         "consumer": f"{mj} {oj}", "wait_states": waits, "exec_partial": partial})
                        break
                    if _regs(_split_operands(oj)[0]) & dregs and not hit:
                        dregs -= _regs(_split_operands(oj)[0])       # overwritten before being read
                        if not dregs:
                            break
                waits += 1
    return findings


def is_guarded(kernel: str) -> bool:
    return any(g in kernel for g in GUARDED)


def check_objects(obj_paths: list, verbose: bool = False) -> dict:
    report = {"pk_min_wait": PK_MIN_WAIT, "kernels": 0, "instructions": 0, "by_rule": {"R1": 0, "R2": 0, "R3": 0},
              "guarded_R1": [], "R1": [], "R2_kernels": {}, "R3_guarded_kernels": {}}
    for obj in obj_paths:
        for name, insns in disassemble_object(obj).items():
            report["kernels"] += 1
            report["instructions"] += sum(1 for x in insns if x[0] != "<label>")
            for f in check_kernel(insns):
                report["by_rule"][f["rule"]] += 1
                f = dict(f, kernel=name, object=os.path.basename(obj))
                if f["rule"] == "R1":
                    report["R1"].append(f)
                    if is_guarded(name):
                        report["guarded_R1"].append(f)
                elif f["rule"] == "R2":
                    report["R2_kernels"][name] = report["R2_kernels"].get(name, 0) + 1
                elif is_guarded(name):
                    report["R3_guarded_kernels"].setdefault(name, []).append(f["text"].split()[0])
    if verbose:
        print(json.dumps({k: v for k, v in report.items() if k not in ("R2_kernels",)}, indent=1)[:4000])
    return report


def check_library(lib_path: str | None = None, verbose: bool = False, enforce: bool = True) -> dict:
    """Check the objects the library was linked from (cbas_amd/build/*.o; the .so's fat binary section is a concatenation
    of their bundles).  Writes asmcheck_report.json; raises if a guarded kernel has an R1 finding."""
    objdir = os.path.join(HERE, "build")
    objs = sorted(os.path.join(objdir, f) for f in os.listdir(objdir) if f.endswith(".o") and not f.endswith(".debug.o"))
    report = check_objects(objs, verbose)
    report["library"] = os.path.basename(lib_path) if lib_path else None
    with open(REPORT, "w") as f:
        json.dump(report, f, indent=1)
    if enforce and report["guarded_R1"]:
        lines = "\n".join(f"  {x['kernel']}: {x['text']}  ->  {x['consumer']}  ({x['wait_states']} wait states, partial EXEC)"
                          for x in report["guarded_R1"][:20])
        raise RuntimeError("asmcheck: packed-fp32 results consumed within "
                           f"{PK_MIN_WAIT} wait state(s) under a partial EXEC in kernels that run beside other work "
                           f"(DESIGN.md section 4):\n{lines}")
    return report


if __name__ == "__main__":
    rep = check_library(verbose="--verbose" in sys.argv or "-v" in sys.argv, enforce="--no-enforce" not in sys.argv)
    print(json.dumps({"kernels": rep["kernels"], "instructions": rep["instructions"], "by_rule": rep["by_rule"],
                      "guarded_R1": len(rep["guarded_R1"]), "R2_kernels": len(rep["R2_kernels"]),
                      "R3_guarded_kernels": sorted(rep["R3_guarded_kernels"])}, indent=1))
