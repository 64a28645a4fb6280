"""The native `_outputs.csv` emitter of the C ABI (cbas_csv_format_f32 / cbas_csv_write_f32, csrc/host_text.cpp) against what
the reference writes: pd.DataFrame(probs, columns=behaviors).to_csv(path, index=False) (backend/cbas.py:565), i.e. numpy's
str(np.float32) per value.  CPU only: the functions touch no GPU."""
import ctypes as C
import io
import os
import time

import numpy as np
import pytest

from cbas_amd import _lib
from cbas_amd.pipeline import csv_header_line, format_probs_csv, write_probs_csv

pd = pytest.importorskip("pandas")
NAMES9 = ["eating", "drinking", "rearing", "climbing", "digging", "nesting", "resting", "grooming", "exploring"]


def _native_text(arr):
    lib = _lib.load()
    arr = np.ascontiguousarray(arr, np.float32)
    n, c = arr.shape
    cap = lib.cbas_csv_format_f32(None, n, c, None, 0)
    buf = C.create_string_buffer(max(1, cap))
    m = lib.cbas_csv_format_f32(arr.ctypes.data, n, c, buf, cap)
    assert m >= 0
    return buf.raw[:m].decode()


def _pandas_text(arr, cols):
    buf = io.StringIO()
    pd.DataFrame(arr, columns=cols).to_csv(buf, index=False)
    return buf.getvalue()


def test_every_exponent_and_boundary_mantissa_equals_numpy():
    ex = np.arange(0, 256, dtype=np.uint32)
    man = np.array([0, 1, 2, 3, 0x400000, 0x7fffff, 0x7ffffe, 0x3fffff, 0x555555, 0x2aaaaa, 0x100000, 0x600000,
                    0x0ccccd, 0x4ccccd, 0x19999a], np.uint32)
    bits = ((ex[:, None] << 23) | man[None, :]).reshape(-1)
    bits = np.concatenate([bits, bits | np.uint32(0x80000000)])
    v = bits.view(np.float32)
    want = ["" if s == "nan" else s for s in v.astype(str).tolist()]          # pandas writes NaN as the empty field
    got = _native_text(v.reshape(-1, 1)).split("\n")[:-1]
    assert got == want


def test_random_bit_patterns_equal_numpy():
    rng = np.random.default_rng(11)
    v = rng.integers(0, 2 ** 32, 400_000, dtype=np.uint64).astype(np.uint32).view(np.float32)
    want = ["" if s == "nan" else s for s in v.astype(str).tolist()]
    got = _native_text(v.reshape(-1, 1)).split("\n")[:-1]
    bad = [(g, w) for g, w in zip(got, want) if g != w]
    assert not bad, bad[:5]


def test_the_positional_threshold_is_numpys():
    # float32(1e-4) is just below 1e-4 as a double -> scientific; 1e16 -> scientific; 9.99e15 -> positional with zeros
    v = np.array([[1e-4, 1.0001e-4, 1e16, 9.99e15, 0.0, -0.0, 1.0, 123456.7, 1e-45, 3.4e38, np.inf, -np.inf, np.nan]], np.float32)
    assert _native_text(v) == "1e-04,0.00010001,1e+16,9990000000000000.0,0.0,-0.0,1.0,123456.7,1e-45,3.4e+38,inf,-inf,\n"
    assert _native_text(v) == _pandas_text(v, list("abcdefghijklm")).split("\n", 1)[1]


def test_written_file_equals_pandas_bytes_single_and_multi_threaded(tmp_path):
    rng = np.random.default_rng(0)
    z = rng.standard_normal((20_011, 9)).astype(np.float32) * 7
    p = np.exp(z - z.max(1, keepdims=True))
    p = (p / p.sum(1, keepdims=True)).astype(np.float32)
    want = _pandas_text(p, NAMES9).encode()
    for threads in (1, 4):
        out = str(tmp_path / f"t{threads}.csv")
        write_probs_csv(out, p, NAMES9, threads=threads)
        assert open(out, "rb").read() == want
    assert format_probs_csv(p[:50], NAMES9) == want.decode().split("\n", 51)[0] + "\n" + "\n".join(want.decode().split("\n")[1:51]) + "\n"
    quoted = ['a,b', 'say "x"', "plain"]
    out = str(tmp_path / "q.csv")
    write_probs_csv(out, p[:3, :3], quoted)
    assert open(out, "rb").read() == _pandas_text(p[:3, :3], quoted).encode()
    assert csv_header_line(quoted) == '"a,b","say ""x""",plain\n'


def test_empty_and_bad_arguments(tmp_path):
    out = str(tmp_path / "e.csv")
    write_probs_csv(out, np.empty((0, 2), np.float32), ["a", "b"])
    assert open(out).read() == "a,b\n"
    with pytest.raises(ValueError):
        write_probs_csv(out, np.zeros((2, 3), np.float32), ["a", "b"])
    with pytest.raises(RuntimeError, match="cannot open"):
        write_probs_csv(str(tmp_path / "no" / "dir.csv"), np.zeros((1, 1), np.float32), ["a"])


def test_throughput_is_far_above_the_pandas_writer(tmp_path):
    """VERDICT r2 asked for >= 300 k rows/s (pandas: ~35 k rows/s at 9 columns)."""
    rng = np.random.default_rng(1)
    p = rng.random((200_000, 9), dtype=np.float32)
    out = str(tmp_path / "big.csv")
    write_probs_csv(out, p[:1000], NAMES9)
    best = 0.0
    for _ in range(3):
        t0 = time.perf_counter()
        write_probs_csv(out, p, NAMES9, threads=1)
        best = max(best, p.shape[0] / (time.perf_counter() - t0))
    print(f"native CSV writer: {best / 1e3:.0f} k rows/s on one thread (9 columns)")
    assert best > 300_000
    assert os.path.getsize(out) > p.shape[0] * 9 * 8
