"""Property tests of the host-side pieces (hypothesis; CPU only).

1. The built-in CSV emitter against pandas itself (backend/cbas.py:565 writes `_outputs.csv` with
pd.DataFrame(probs, columns=behaviors).to_csv(index=False)): random float32 bit patterns - subnormals, values that
need 9 significant digits, exact powers of two, negative zero - and header names that need quoting.
2. Clip / frame sharding (cbas_amd/dist.py): every unit owned exactly once, ranges contiguous and balanced, the
   interleave is the inverse of the round-robin assignment.
3. The head's window semantics on the CPU oracle: classifying a clip range by range with the +-half halo equals
   classifying it whole (the property infer_file's 20 000-frame chunks and the frame-sharded multi-GPU path rely on)."""
import io

import numpy as np
import pytest
from hypothesis import given, settings, strategies as st

pd = pytest.importorskip("pandas")
from cbas_amd.pipeline import format_probs_csv

finite_f32 = st.integers(0, 2 ** 32 - 1).map(lambda b: np.uint32(b).view(np.float32)).filter(np.isfinite)
names = st.text(alphabet=st.sampled_from(list("abcXYZ 019_-,\"'")), min_size=1, max_size=8)


def _pandas_text(arr, cols):
    buf = io.StringIO()
    pd.DataFrame(arr, columns=cols).to_csv(buf, index=False)
    return buf.getvalue()


@settings(max_examples=200, deadline=None)
@given(st.lists(finite_f32, min_size=1, max_size=24), st.integers(1, 4))
def test_float32_text_equals_pandas(values, ncol):
    n = len(values) // ncol * ncol
    if n == 0:
        values, n, ncol = values[:1], 1, 1
    arr = np.array(values[:n], np.float32).reshape(-1, ncol)
    cols = [f"c{i}" for i in range(ncol)]
    assert format_probs_csv(arr, cols) == _pandas_text(arr, cols)


@settings(max_examples=100, deadline=None)
@given(st.lists(names, min_size=1, max_size=5, unique=True))
def test_header_quoting_equals_pandas(cols):
    arr = np.linspace(0.0, 1.0, 2 * len(cols), dtype=np.float32).reshape(2, -1)
    assert format_probs_csv(arr, cols) == _pandas_text(arr, cols)


def test_softmax_like_rows_equal_pandas():
    rng = np.random.default_rng(0)
    z = rng.standard_normal((2000, 9)).astype(np.float32) * 6
    p = np.exp(z - z.max(1, keepdims=True))
    p = (p / p.sum(1, keepdims=True)).astype(np.float32)
    cols = ["eating", "drinking", "rearing", "climbing", "digging", "nesting", "resting", "grooming", "exploring"]
    assert format_probs_csv(p, cols) == _pandas_text(p, cols)


@settings(max_examples=200, deadline=None)
@given(st.integers(0, 300), st.integers(1, 16))
def test_clip_sharding_is_a_partition_and_interleave_inverts_it(n_clips, world):
    import torch
    from cbas_amd import dist as D
    owned = [D.shard_clips(n_clips, world, r) for r in range(world)]
    flat = sorted(c for o in owned for c in o)
    assert flat == list(range(n_clips))
    assert all(D.owner_of(c, world) == r for r, o in enumerate(owned) for c in o)
    assert max(len(o) for o in owned) - min(len(o) for o in owned) <= 1
    per_rank = [[torch.full((1, 1), float(c)) for c in o] for o in owned]
    back = D.interleave_by_clip(per_rank, n_clips)
    assert [int(t.item()) for t in back] == list(range(n_clips))


@settings(max_examples=200, deadline=None)
@given(st.integers(0, 100000), st.integers(1, 16))
def test_frame_sharding_is_contiguous_and_balanced(n_frames, world):
    from cbas_amd import dist as D
    ranges = [D.shard_frames(n_frames, world, r) for r in range(world)]
    assert ranges[0][0] == 0 and ranges[-1][1] == n_frames
    assert all(ranges[r][1] == ranges[r + 1][0] for r in range(world - 1))
    sizes = [b - a for a, b in ranges]
    assert min(sizes) >= 0 and max(sizes) - min(sizes) <= 1


@settings(max_examples=10, deadline=None)
@given(st.integers(1, 70), st.sampled_from([4, 9, 16, 33]), st.sampled_from([3, 512]), st.sampled_from([7, 15, 31]))
def test_oracle_halo_chunks_do_not_change_the_result(n, chunk, batch, seq_len):
    """backend/cbas.py:497-551 reads [start-half, end+half) per 20 000-frame chunk and pads only at the video's ends: the
    chunk size (and the 512-window batch) must not show in the output - here on the CPU restatement, with small chunks."""
    from cbas_amd import config as C, weights as W, synth
    from oracle import head_oracle as HO
    hc = C.HeadConfig(in_features=64, out_features=5, seq_len=seq_len, lstm_hidden_size=16, bottleneck_dim=16, lin0_dim=32)
    w = W.synth_head_weights(hc, 7)
    rows = synth.cls_walk(3, n, 64).astype(np.float16)
    whole = HO.infer_file_literal(rows, w, seq_len, 1.0)
    got = HO.infer_file_literal(rows, w, seq_len, 1.0, chunk=chunk, batch=batch)
    assert got.shape == (n, 5) and np.allclose(got, whole, atol=1e-6)
