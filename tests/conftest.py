import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: CPU test that takes more than ~20 s")


def _have_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _have_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


def assert_labels_match(probs, ref_probs, prob_tol):
    """Argmax-label parity, stated precisely: the probabilities agree within ``prob_tol`` and every
    frame's label is identical except where the REFERENCE itself is within 2*|dp|max of a tie
    between its top two classes (a flip there is implied by any non-bit-exact arithmetic, the
    reference's own fp16 GPU path included).  Returns (n_mismatch, n_near_tie_frames)."""
    import numpy as np
    probs, ref_probs = np.asarray(probs, np.float64), np.asarray(ref_probs, np.float64)
    dp = float(np.abs(probs - ref_probs).max())
    assert dp <= prob_tol, f"probabilities differ by {dp:.3e} > {prob_tol:.1e}"
    s = np.sort(ref_probs, axis=1)
    margin = s[:, -1] - s[:, -2] if ref_probs.shape[1] > 1 else np.ones(len(ref_probs))
    mism = probs.argmax(1) != ref_probs.argmax(1)
    near = margin <= 2.0 * dp
    # the relaxation is always visible in the test log (-s): how many labels moved and how wide the band was
    print(f"[labels] {int(mism.sum())} of {len(probs)} argmax labels differ; |dp|max = {dp:.3e}, near-tie band "
          f"(reference top-2 margin <= {2.0 * dp:.3e}) holds {int(near.sum())} frames; flips outside the band: "
          f"{int(np.sum(mism & ~near))}")
    assert not np.any(mism & ~near), f"label flips outside the near-tie band: frames {np.nonzero(mism & ~near)[0][:10]}"
    return int(mism.sum()), int(near.sum())
