"""Root cause of round 4's co-residency corruption: instruction-level A/B of the OLD head_expand_kernel.

Round 4 found the head's expand kernel (commit 93483c0) returning wrong values in lanes 48-63 of its third stream when waves
of another kernel kept the matrix pipe busy on the same CU, removed it empirically, and left the cause open.  This script
rebuilds that kernel (scripts/probes/expand_r4/expand_r4.hip: the same instruction stream, checked) as stand-alone gfx950
code objects - verbatim, with source-level changes, and with single edits of its ASSEMBLY - and runs each IN PLACE of the
library's kernel inside the real head pipeline (cbas_head_debug_expand_module, debug build) beside the register-only
v_mfma_f32_32x32x16_f16 loop (cbas_debug_mfma_neighbor).  For every differing run it reads the head's workspace back, finds the
wrong LayerNorm input values, inverts LayerNorm + GELU on the host and regresses the recovered pre-activation value on the
kernel's own operands (a, b, c = EMA at t, t-1, t-2): the coefficients say WHICH register was read stale.

    python scripts/expand_rootcause.py build                      # code objects -> scripts/probes/bin/expand_r4_*.hsaco
    python scripts/expand_rootcause.py run [seconds [out.json]]   # on the GPU box: one pass per variant, no retries

Variants (the third stream's value is v = (a - b) - (b - c); in the old code: ds_read a, b, c; s_and_saveexec <stream 2>;
s_waitcnt lgkmcnt(0); v_pk_add_f32 {b - c, a - b}; s_nop 0; v_sub_f32; ...; divergent erff):
  r4            verbatim
  nopk          source: compiler barrier between the two differences (no v_pk_add_f32; divergent erff and LDS read-back kept)
  erfbf         source: branch-free erf (v_pk_add_f32 under partial EXEC and LDS read-back kept)   [the converse]
  nop_after_pk  asm: `s_nop 0` after the v_pk_add_f32 -> `s_nop 7`     (consumer 8 wait states away)
  scalar_subs   asm: v_pk_add_f32 + v_sub_f32 -> three v_sub_f32, everything else untouched
  nop_before_pk asm: `s_nop 7` between `s_waitcnt lgkmcnt(0)` and the v_pk_add_f32   (LDS data given 8 more wait states)
  wait_early    asm: the three ds_reads waited for (lgkmcnt(0)) BEFORE EXEC is narrowed, then `s_nop 7`
  pk_plain      asm: the same packed subtraction WITHOUT the cross-half operand selection and without dst = src1: c and b are
                moved into a fresh pair first, `v_pk_add_f32 v[8:9], v[10:11], v[18:19] neg_lo neg_hi` (the form the current
                library kernel compiles to)
  pk_nooverlap  asm: the cross-half selection kept, result into a fresh pair (v[18:19]) instead of over src1
  pk_mov        asm: the cross-half selection in a MOVE (v_pk_mov_b32 ... op_sel:[1,0] builds {c, b}), the subtraction itself without
                operand selection: arithmetic or operand routing?
  pk_bcast      asm: the MIRROR form: {a - b, c - b} with b broadcast from the low register of its pair (op_sel_hi:[1,0]), then
                (a - b) + (c - b): the scalar-broadcast form the compiler uses throughout the GEMM epilogues (asmcheck R2)

Amplification (cbas_head_debug_expand_repeat): every pass launches the probe kernel R times and a device-side kernel compares
each launch's rows with a reference taken on the idle device, capturing the differing rows: ~60 x the launches per second of
the plain pipeline, so that per-variant counts mean something on a box where the fault is rare.
"""
import json, os, subprocess, sys, threading, time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PROBE = os.path.join(REPO, "scripts", "probes", "expand_r4", "expand_r4.hip")
BIN = os.path.join(REPO, "scripts", "probes", "bin")
LLVM = "/opt/rocm/lib/llvm/bin"
KERNEL = "head_expand_r4"
PK = "v_pk_add_f32 v[8:9], v[10:11], v[8:9] op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]"


def _asm(defs=()):
    out = subprocess.run(["hipcc", "-S", "--cuda-device-only", "--offload-arch=gfx950", "-O3", "-std=c++17", "-I",
                          os.path.join(REPO, "cbas_amd", "csrc"), *[f"-D{d}" for d in defs], PROBE, "-o", "-"],
                         capture_output=True, text=True)
    if out.returncode != 0:
        raise RuntimeError(out.stderr)
    return out.stdout


def _hsaco(name, asm):
    os.makedirs(BIN, exist_ok=True)
    s, o, h = (os.path.join(BIN, f"expand_r4_{name}.{e}") for e in ("s", "o", "hsaco"))
    open(s, "w").write(asm)
    subprocess.run([f"{LLVM}/clang", "-x", "assembler", "-target", "amdgcn-amd-amdhsa", "-mcpu=gfx950", "-c", s, "-o", o], check=True)
    subprocess.run([f"{LLVM}/ld.lld", "-shared", o, "-o", h], check=True)
    os.remove(o)
    return h


def _edit(asm, old, new, count=1):
    assert asm.count(old) == count, (old, asm.count(old))
    return asm.replace(old, new)


def build():
    base = _asm()
    assert base.count("v_pk_") == 1 and PK in base, "the probe no longer compiles to the round-4 instruction stream"
    seq = f"\ts_waitcnt lgkmcnt(0)\n\t{PK}\n\ts_nop 0\n\tv_sub_f32_e32 v18, v9, v8\n"
    assert base.count(seq) == 1
    variants = {"r4": base}
    nopk = _asm(["EXPAND_NOPK=1"])
    assert "v_pk_" not in nopk, "EXPAND_NOPK still pairs the differences"
    variants["nopk"] = nopk
    variants["erfbf"] = _asm(["EXPAND_ERF=1"])
    assert "v_pk_add_f32" in variants["erfbf"]
    variants["nop_after_pk"] = _edit(base, seq, f"\ts_waitcnt lgkmcnt(0)\n\t{PK}\n\ts_nop 7\n\tv_sub_f32_e32 v18, v9, v8\n")
    # lo = b - c = v10 - v9, hi = a - b = v11 - v8 ; v18 = hi - lo.  v18 is dead here (overwritten by this very sequence).
    variants["scalar_subs"] = _edit(base, seq, "\ts_waitcnt lgkmcnt(0)\n\tv_sub_f32_e32 v18, v11, v8\n\tv_sub_f32_e32 v8, v10, v9\n"
                                               "\ts_nop 0\n\tv_sub_f32_e32 v18, v18, v8\n")
    variants["nop_before_pk"] = _edit(base, seq, f"\ts_waitcnt lgkmcnt(0)\n\ts_nop 7\n\t{PK}\n\ts_nop 0\n\tv_sub_f32_e32 v18, v9, v8\n")
    reads = ("\tds_read_b32 v10, v8\n\tds_read_b32 v11, v17\n\tds_read_b32 v9, v9\n\ts_waitcnt lgkmcnt(2)\n\tv_mov_b32_e32 v8, v10\n"
             "\ts_waitcnt lgkmcnt(1)\n\tv_mov_b32_e32 v18, v11\n")
    variants["wait_early"] = _edit(base, reads, "\tds_read_b32 v10, v8\n\tds_read_b32 v11, v17\n\tds_read_b32 v9, v9\n\ts_waitcnt lgkmcnt(0)\n"
                                                "\ts_nop 7\n\tv_mov_b32_e32 v8, v10\n\tv_mov_b32_e32 v18, v11\n")
    variants["pk_plain"] = _edit(base, seq, "\ts_waitcnt lgkmcnt(0)\n\tv_mov_b32_e32 v18, v9\n\tv_mov_b32_e32 v19, v8\n"
                                            "\tv_pk_add_f32 v[8:9], v[10:11], v[18:19] neg_lo:[0,1] neg_hi:[0,1]\n\ts_nop 0\n"
                                            "\tv_sub_f32_e32 v18, v9, v8\n")
    variants["pk_nooverlap"] = _edit(base, seq, "\ts_waitcnt lgkmcnt(0)\n\tv_pk_add_f32 v[18:19], v[10:11], v[8:9] op_sel:[0,1] op_sel_hi:[1,0] "
                                                "neg_lo:[0,1] neg_hi:[0,1]\n\ts_nop 0\n\tv_sub_f32_e32 v18, v19, v18\n")
    # the mirror form (asmcheck R2): HIGH halves... {a - b, c - b} with b broadcast from the LOW register of its pair
    # (op_sel_hi:[1,0]); (a - b) + (c - b) is bit for bit (a - b) - (b - c)
    variants["pk_bcast"] = _edit(base, seq, "\ts_waitcnt lgkmcnt(0)\n\tv_mov_b32_e32 v18, v11\n\tv_mov_b32_e32 v19, v9\n"
                                            "\tv_pk_add_f32 v[18:19], v[18:19], v[8:9] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n\ts_nop 0\n"
                                            "\tv_add_f32_e32 v18, v18, v19\n")
    # the same cross-half selection in a MOVE: {c, b} = v_pk_mov_b32 of {b, c} with op_sel:[1,0] (low result = the source's HIGH
    # register), then the subtraction without any operand selection - is it the arithmetic or the operand routing?
    variants["pk_mov"] = _edit(base, seq, "\ts_waitcnt lgkmcnt(0)\n\tv_pk_mov_b32 v[18:19], v[8:9], v[8:9] op_sel:[1,0]\n\ts_nop 0\n"
                                          "\tv_pk_add_f32 v[8:9], v[10:11], v[18:19] neg_lo:[0,1] neg_hi:[0,1]\n\ts_nop 0\n"
                                          "\tv_sub_f32_e32 v18, v9, v8\n")
    # v18 / v19 are dead at the edited point: v18 is written by this very sequence, v19 only later (a temporary inside erff)
    return {k: _hsaco(k, v) for k, v in variants.items()}


# ---- host restatement of the expand kernel's first two passes (what each lane should have computed) --------------------
def _gelu(x):
    import numpy as np
    from scipy.special import erf
    x = np.asarray(x, np.float64)
    return 0.5 * x * (1.0 + erf(x * 0.70710678118654752440))


def _inv_gelu(g, hint):
    """Pre-activation x with gelu(x) = g nearest to `hint` (gelu is not monotonic below ~ -0.75: Newton from the hint)."""
    import numpy as np
    from scipy.special import erf
    x = np.array(hint, np.float64)
    for _ in range(60):
        f = _gelu(x) - g
        d = 0.5 * (1.0 + erf(x / np.sqrt(2))) + x * np.exp(-0.5 * x * x) / np.sqrt(2 * np.pi)
        x = x - f / np.where(np.abs(d) < 1e-6, 1e-6, d)
    return x


def analyse_row(proj, row_index, y, y_ref, dims, b_bott, ln_w, ln_b, n_frames):
    """One differing LayerNorm OUTPUT row (y, against the idle device's y_ref) -> which lanes held a wrong LayerNorm input,
    what that input was (LayerNorm inverted through the row's good lanes), the candidate formula it equals, and a regression of
    the recovered pre-activation value on the kernel's own operands a, b, c = EMA at t, t-1, t-2."""
    import numpy as np
    T, NS, Bn, alpha = dims["T"], dims["NS"], dims["Bn"], np.float32(dims["alpha"])
    half = T // 2
    st = int(row_index % NS); t = int((row_index // NS) % T); w = int(row_index // (NS * T))
    frames = np.clip(w + np.arange(T) - half, 0, n_frames - 1)
    x = proj[frames][:, st * Bn:(st + 1) * Bn].astype(np.float32)                 # [T][Bn] this stream's projected rows
    s = np.empty_like(x)
    s[0] = x[0]
    for k in range(1, T):                                                          # v_fmac: fma(alpha, x - s, s)
        s[k] = (s[k - 1].astype(np.float64) + np.float64(alpha) * (x[k] - s[k - 1]).astype(np.float64)).astype(np.float32)
    pad = np.concatenate([s[2:3], s[1:2], s], 0)                                   # reflect padding [s2, s1 | s0, s1, ...]
    a, b, c = pad[t + 2], pad[t + 1], pad[t]
    true_v = {0: a, 1: a - b, 2: (a - b) - (b - c)}[st]
    bb, lw, lb = (v[st * Bn:(st + 1) * Bn].astype(np.float64) for v in (b_bott, ln_w, ln_b))
    g_true = _gelu(true_v.astype(np.float64) + bb)
    y, y_ref = np.asarray(y, np.float64), np.asarray(y_ref, np.float64)
    ok_w = np.abs(lw) > 1e-3
    z = np.where(ok_w, (y - lb) / np.where(ok_w, lw, 1), np.nan)                   # = (g - mean) * rstd
    good = ok_w.copy()
    for _ in range(4):                                                             # robust line z = A g + B
        A, B = np.polyfit(g_true[good], z[good], 1)
        res = np.abs(z - (A * g_true + B))
        good = ok_w & (res < max(1e-4, 20 * np.nanmedian(res[good])))
    wrong = np.nonzero(ok_w & ~good)[0]
    g_rec = (z[wrong] - B) / A
    cand = {"(a-b)-(b-c) [true]": true_v, "a": a, "a-b": a - b, "b-c": b - c, "c-b [v_sub read both halves stale]": c - b,
            "c-(b-c) [v_sub read stale hi]": c - (b - c), "(a-b)-b [lo half = b: stale lo in v_sub, or c read as 0]": (a - b) - b,
            "-(b-c)": -(b - c), "b": b, "c": c, "-b": -b, "a-(b-c)-(b-c) [hi formed from the new lo]": a - 2 * b + 2 * c,
            "(a-b)-(b-(a-b)) [lo formed from the new hi]": 2 * a - 3 * b, "0": np.zeros_like(a)}
    match = {}
    for name, v in cand.items():
        gc = _gelu(v.astype(np.float64)[wrong] + bb[wrong])
        match[name] = float(np.max(np.abs(gc - g_rec))) if len(wrong) else None
    best = min((v, k) for k, v in match.items() if v is not None) if len(wrong) else (None, None)
    coef = None
    if len(wrong) >= 6:
        guess = cand[best[1]].astype(np.float64)[wrong] + bb[wrong] if best[0] is not None and best[0] < 1e-3 else true_v.astype(np.float64)[wrong] + bb[wrong]
        v_rec = _inv_gelu(g_rec, guess) - bb[wrong]
        M = np.stack([a[wrong], b[wrong], c[wrong], np.ones(len(wrong))], 1).astype(np.float64)
        sol, *_ = np.linalg.lstsq(M, v_rec, rcond=None)
        coef = {"a": round(float(sol[0]), 4), "b": round(float(sol[1]), 4), "c": round(float(sol[2]), 4), "const": round(float(sol[3]), 4),
                "max_resid": float(np.max(np.abs(M @ sol - v_rec)))}
    return {"window": w, "t": t, "stream": st, "wrong_lanes": wrong.tolist(), "fit_residual_good_lanes": float(np.nanmax(res[good])),
            "best_candidate": best[1], "best_candidate_err": best[0], "regression_v_on_a_b_c": coef,
            "candidate_max_abs_err": {k: (None if v is None else float(f"{v:.3e}")) for k, v in match.items()}}


def run(seconds: float, out_path, repeat: int = 64):
    os.environ["CBAS_BUILD_DEBUG"] = "1"
    sys.path.insert(0, REPO)
    import ctypes as C
    from collections import Counter
    import numpy as np, torch
    from cbas_amd import config as Cfg, weights as W, _lib
    from cbas_amd.head import ClassifierLSTMDeltas
    names = tuple(os.environ.get("EXPAND_VARIANTS", "r4,nopk,erfbf,nop_after_pk,scalar_subs,nop_before_pk,wait_early,pk_plain,pk_nooverlap,pk_bcast,pk_mov").split(","))
    paths = {k: os.path.join(BIN, f"expand_r4_{k}.hsaco") for k in names}
    if not all(os.path.exists(p) for p in paths.values()):
        built = build()
        paths = {k: built[k] for k in names}
    lib = _lib.load()
    hc = Cfg.HeadConfig()
    sd = W.synth_head_weights(hc, 4321)
    head = ClassifierLSTMDeltas(768, 9)
    head.load_state_dict(sd)
    head.to("cuda")
    n = 3000
    rows = torch.from_numpy(np.random.default_rng(5).standard_normal((n, 768)).astype(np.float16)).cuda()
    lib_ref = head.infer_clip(rows).clone()
    torch.cuda.synchronize()
    T, NS, Bn, Cc = hc.seq_len, 3, hc.bottleneck_dim, 9
    dims = {"T": T, "NS": NS, "Bn": Bn, "NPROJ": (NS * Bn + Cc + 3) // 4 * 4, "alpha": hc.ema_alpha}
    b_bott = np.concatenate([sd[f"{k}_bottleneck.0.bias"] for k in ("cls", "delta", "acc")]).astype(np.float32)
    ln_w = np.concatenate([sd[f"{k}_ln.weight"] for k in ("cls", "delta", "acc")]).astype(np.float32)
    ln_b = np.concatenate([sd[f"{k}_ln.bias"] for k in ("cls", "delta", "acc")]).astype(np.float32)

    def read(which, count):
        buf = np.empty(count, np.float32)
        _lib.check(lib.cbas_head_debug_read(head._h, which, buf.ctypes.data, count), "cbas_head_debug_read")
        return buf

    proj = read(1, n * dims["NPROJ"]).reshape(n, dims["NPROJ"]).copy()      # the same rows every pass: the projection is constant
    s2 = torch.cuda.Stream()
    dev = torch.cuda.current_device()
    results = {"seconds_per_variant": seconds, "launches_per_pass": repeat, "windows_per_launch": n,
               "neighbour": "cbas_debug_mfma_neighbor(20000) on the default stream", "variants": {}}
    for name in ["library"] + list(paths) + (["r4_again"] if "r4" in paths else []):
        key = "r4" if name == "r4_again" else name
        if key == "library":
            _lib.check(lib.cbas_head_debug_expand_module(head._h, None, None), "expand_module")
        else:
            _lib.check(lib.cbas_head_debug_expand_module(head._h, paths[key].encode(), KERNEL.encode()), "expand_module")
        _lib.check(lib.cbas_head_debug_expand_repeat(head._h, 1, 1), "expand_repeat")          # idle pass -> device reference
        ref = head.infer_clip(rows).clone()
        torch.cuda.synchronize()
        aug_ref = read(2, n * T * NS * Bn).reshape(n * T * NS, Bn).copy()
        _lib.check(lib.cbas_head_debug_expand_repeat(head._h, 2, repeat), "expand_repeat")
        cnt = (C.c_uint64 * 4)()
        _lib.check(lib.cbas_head_debug_expand_stats(head._h, cnt, None, 0, 1), "expand_stats")  # clear
        rec = {"idle_equals_library_kernel": bool(torch.equal(ref, lib_ref)), "head_passes": 0, "head_passes_differing": 0}
        stop = threading.Event()

        def head_loop():
            torch.cuda.set_device(dev)
            with torch.cuda.stream(s2):
                while not stop.is_set():
                    o = head.infer_clip(rows)
                    s2.synchronize()
                    rec["head_passes"] += 1
                    rec["head_passes_differing"] += int(not torch.equal(o, ref))
        th = threading.Thread(target=head_loop)
        th.start()
        t0, k = time.time(), 0
        try:
            while time.time() - t0 < seconds:
                _lib.check(lib.cbas_debug_mfma_neighbor(20000, None), "mfma_neighbor")
                torch.cuda.synchronize()
                k += 1
        finally:
            stop.set()
            th.join()
        cap = np.empty((512, 2 + Bn), np.float32)
        _lib.check(lib.cbas_head_debug_expand_stats(head._h, cnt, cap.ctypes.data, 512, 1), "expand_stats")
        rec.update(neighbour_launches=k, expand_launches=int(cnt[0]), expand_launches_differing=int(cnt[1]), rows_differing=int(cnt[2]))
        got = cap[: min(int(cnt[3]), 512)]
        rows_an = [analyse_row(proj, int(r[0]), r[2:], aug_ref[int(r[0])], dims, b_bott, ln_w, ln_b, n) for r in got[:48]]
        rec["rows_analysed"] = len(rows_an)
        rec["streams"] = dict(Counter(str(r["stream"]) for r in rows_an))
        rec["wrong_lane_sets"] = dict(Counter(f"{min(r['wrong_lanes'])}-{max(r['wrong_lanes'])} ({len(r['wrong_lanes'])})" if r["wrong_lanes"] else "none"
                                              for r in rows_an))
        rec["best_candidates"] = dict(Counter(f"{r['best_candidate']}" if (r["best_candidate_err"] or 1) < 1e-4 else "no candidate matches" for r in rows_an))
        rec["examples"] = rows_an[:3]
        results["variants"][name] = rec
        print(json.dumps({name: {kk: vv for kk, vv in rec.items() if kk != "examples"}}), flush=True)
        if out_path:
            json.dump(results, open(out_path, "w"), indent=1)
    _lib.check(lib.cbas_head_debug_expand_repeat(head._h, 0, 1), "expand_repeat")
    _lib.check(lib.cbas_head_debug_expand_module(head._h, None, None), "expand_module")
    head.close()
    return results


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "build":
        for k, v in build().items():
            print(k, v)
    else:
        secs = float(sys.argv[2]) if len(sys.argv) > 2 else 40.0
        run(secs, sys.argv[3] if len(sys.argv) > 3 else None, int(sys.argv[4]) if len(sys.argv) > 4 else 64)
