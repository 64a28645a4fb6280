"""`python bench.py --gpus N` must produce an N-rank run by itself: with N > 1 and no launcher around it,
bench.py starts the ranks as child processes; it never measures one GPU under an N-GPU label."""
import json
import os
import subprocess
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(argv, env_extra=None, timeout=900):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "CBAS_BUILD_DEBUG")}
    env.update(env_extra or {})
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), *argv], env=env, capture_output=True, text=True, timeout=timeout)
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    return r, lines


def test_more_ranks_than_devices_over_rccl_is_refused_with_a_null_value():
    """Here (no GPU) and on a one-GPU box alike: `--gpus 2` over RCCL cannot run; the single line says so and the exit code is 2."""
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("two devices visible: the request is satisfiable")
    r, lines = _run(["--gpus", "2", "--steps", "3", "--warmup", "1"], timeout=300)
    assert r.returncode == 2, (r.returncode, r.stderr[-2000:])
    assert len(lines) == 1
    rec = json.loads(lines[0])
    assert rec["value"] is None and rec["n_gpus"] == 2 and rec["ranks_seen"] == 0 and "device(s) are visible" in rec["error"]


@pytest.mark.gpu
def test_gpus_2_starts_two_ranks_by_itself_gloo_rehearsal():
    """CBAS_DIST_BACKEND=gloo python bench.py --gpus 2: two child ranks (sharing this box's GPU - a rehearsal of the control
    flow, not a measurement), ONE line on stdout, n_gpus = ranks_seen = 2, weak scaling: 2 x K x batch frames."""
    r, lines = _run(["--gpus", "2", "--steps", "6", "--warmup", "2", "--no-cpu-baseline", "--no-gates", "--files", "1", "--clip-frames", "512",
                     "--preroll-seconds", "0.2"], {"CBAS_DIST_BACKEND": "gloo"})
    assert r.returncode == 0, (r.returncode, r.stdout[-1000:], r.stderr[-3000:])
    assert len(lines) == 1, lines
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["ranks_seen"] == 2 and rec["scaling"] == "weak"
    assert rec["value"] and rec["value"] > 0
    assert rec["config"]["frames_per_gpu"] == 6 * 64
    assert abs(rec["value"] - 2 * 6 * 64 / (rec["ms_per_step"] * 6e-3)) < 0.01 * rec["value"]
    assert rec["files_path"]["value"]


@pytest.mark.gpu
def test_default_line_carries_the_contract_fields():
    """One JSON line on stdout with the fields the driver and the judge read: BASELINE.json's metric string verbatim, whole-job
    frames/s, `roofline` (bound, achieved, peak, frac, traffic) from HIP events taken live, `cpu_baseline` (value, cores, kind,
    sample) with the ViT-S cfg1 figure beside ViT-B's, and the label statement next to `value` (fp16 operands: not identical)
    and inside `label_exact` (precision 4, the drop-in's default: identical).  Runs on the PRODUCT library (no debug symbols)."""
    r, lines = _run(["--steps", "12", "--warmup", "2", "--cpu-frames", "64", "--files", "1", "--clip-frames", "512", "--preroll-seconds", "0.2"],
                    timeout=900)
    assert r.returncode == 0, (r.returncode, r.stderr[-3000:])
    assert len(lines) == 1, lines
    rec = json.loads(lines[0])
    base = json.load(open(os.path.join(REPO, "BASELINE.json")))
    assert rec["metric"] == base["metric"] and rec["unit"] == "frames/s" and rec["higher_is_better"] is True
    assert rec["n_gpus"] == 1 and rec["ranks_seen"] == 1 and rec["steps"] == 12 and rec["warmup"] == 2 and rec["scaling"] == "weak"
    assert rec["vs_baseline"] is None and rec["data"] == "synthetic" and rec["dtype"] == "f16"
    assert abs(rec["value"] - 12 * 64 / (rec["ms_per_step"] * 12e-3)) < 0.01 * rec["value"] and rec["value"] > 2000
    rf = rec["roofline"]
    assert rf["bound"] == "mfma" and rf["unit"] == "TFLOP/s" and rf["peak"] == 2500.0
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3 and 0.1 < rf["frac"] < 1.0 and "traffic" in rf
    cb = rec["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0 and "sample" in cb
    assert cb["cfg1_vits16"]["value"] > cb["value"]                       # ViT-S/16 (cfg1) beside ViT-B/16
    assert rec["labels_identical"] is False                                # the fp16-operand mode flips near-tie labels ...
    le = rec["label_exact"]
    assert le["precision"] == 4 and le["labels_identical"] is True and le["value"] > 2000      # ... precision 4 does not
    assert rec["gates"]["cls_rel_err_max"] <= 1e-3 and le["gates"]["cls_rel_err_max"] <= 5e-6
    assert rec["value_r3_definition"]["value"] > 0 and rec["hbm_resident"]["bit_identical_to_value_pass"] is True
