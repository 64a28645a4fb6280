"""RCCL on the one GPU this pool offers (scripts/rccl_first_contact.py, in its own process so that no process group outlives
it): a world of one rank on the "nccl" backend through cbas_amd.dist's own helpers - communicator creation, barrier, the
all_reduce / all_gather of the control steps, and the grouped point-to-point transfer of a clip's rows posted from a second
thread while the encoder runs on its streams, bytes compared.  Two ranks cannot share a GPU on RCCL ("Duplicate GPU detected",
tried on this pool in round 4), so xGMI, IPC handles and a second process stay untested until a multi-GPU node runs
tests/test_zz_rccl_two_gpus.py."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_rccl_single_rank_first_contact(tmp_path):
    out = tmp_path / "rccl.json"
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29533")
    try:
        r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "rccl_first_contact.py"), str(out)], env=env, cwd=ROOT,
                           capture_output=True, text=True, timeout=240)
    except subprocess.TimeoutExpired as e:                       # a hang here must read as a failure, not stall the suite
        pytest.fail(f"RCCL first contact did not finish in 240 s; output so far:\n{(e.stdout or b'')[-2000:]}")
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    doc = json.loads(out.read_text())
    assert doc["ok"] is True
    steps = {s["step"].split(" (")[0]: s["result"] for s in doc["steps"]}
    assert steps["all_reduce MAX"] == 3.25
    assert steps["all_gather of a row-count block"] == [2, 768, 0, 4096, 17]
    p2p = [s["result"] for s in doc["steps"] if s["step"].startswith("grouped isend")][0]
    assert p2p == {"p2p": "ok", "bytes_identical": True}
