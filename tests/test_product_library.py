"""The PRODUCT build of the library (libcbas_mi355x.so: exactly include/cbas_mi355x.h) on the GPU, in its own process.

The rest of the GPU suite runs on the debug build (tests/conftest.py sets CBAS_BUILD_DEBUG=1 because many tests use stage
taps and harnesses); this file runs __graft_entry__.smoke() - tiny ViT + head against the oracle in precisions 0, 3 and 4 -
in a child process WITHOUT that variable and checks which shared object the child mapped."""
import os
import subprocess
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import os, sys
sys.path.insert(0, {repo!r})
assert os.environ.get("CBAS_BUILD_DEBUG", "0") in ("", "0")
import __graft_entry__ as g
g.smoke()
from cbas_amd import _lib
maps = open("/proc/self/maps").read()
assert "libcbas_mi355x.so" in maps and "libcbas_mi355x_debug.so" not in maps, "wrong library mapped"
assert not hasattr(_lib.load(), "cbas_debug_mfma_neighbor")
try:
    from cbas_amd.encoder import DinoEncoder
    from cbas_amd import config as C, weights as W
    enc = DinoEncoder.from_weights(C.VIT_TINY, W.synth_encoder_weights(C.VIT_TINY, 1), "cuda:0", max_batch=8, max_frame=(64, 64))
    enc.debug_option("rope_lds", 0)
    raise SystemExit("debug_option worked on the product build")
except RuntimeError as e:
    assert "CBAS_BUILD_DEBUG" in str(e), e
print("__PRODUCT_OK__")
"""


@pytest.mark.gpu
def test_product_library_runs_smoke_in_its_own_process():
    env = {k: v for k, v in os.environ.items() if k != "CBAS_BUILD_DEBUG"}
    r = subprocess.run([sys.executable, "-c", CHILD.format(repo=REPO)], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "__PRODUCT_OK__" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
