"""ctypes binding of libcbas_mi355x.so (declared in include/cbas_mi355x.h).

With CBAS_BUILD_DEBUG=1 in the environment the DEBUG build is loaded instead (libcbas_mi355x_debug.so: the same library
plus the bring-up / test entry points of include/cbas_mi355x_debug.h).  The GPU test suite and scripts/ set it; the
product library exports none of those symbols and `require_debug()` says so when one is asked for.

There is no CPU fallback: if the library is missing it is built with hipcc; if that fails, or a
call returns a non-zero code, a ``RuntimeError`` carrying ``cbas_last_error()`` is raised.
"""
from __future__ import annotations

import ctypes as C
import os
import threading

from . import build as _build

_lock = threading.Lock()
_lib = None

c_void_p, c_int, c_int32, c_int64, c_float = C.c_void_p, C.c_int, C.c_int32, C.c_int64, C.c_float


class EncConfig(C.Structure):
    _fields_ = [("hidden_size", c_int32), ("intermediate_size", c_int32), ("num_layers", c_int32),
                ("num_heads", c_int32), ("num_register_tokens", c_int32), ("patch_size", c_int32),
                ("layer_norm_eps", c_float), ("rope_theta", c_float), ("max_batch", c_int32),
                ("max_height", c_int32), ("max_width", c_int32), ("precision", c_int32),
                ("use_rope", c_int32), ("pos_embed_grid", c_int32)]


class HeadConfigC(C.Structure):
    _fields_ = [("in_features", c_int32), ("out_features", c_int32), ("seq_len", c_int32),
                ("bottleneck_dim", c_int32), ("lin0_dim", c_int32), ("lstm_hidden_size", c_int32),
                ("center_window_size", c_int32), ("ema_alpha", c_float), ("lstm_layers", c_int32),
                ("use_acceleration", c_int32)]


class TrainConfigC(C.Structure):
    _fields_ = [("lr", c_float), ("weight_decay", c_float), ("label_smoothing", c_float), ("max_batch", c_int32),
                ("seed", C.c_uint64), ("dropout", c_int32)]


# name -> (restype, argtypes); every symbol include/cbas_mi355x.h declares
SIGNATURES = {
    "cbas_enc_weights_count": (c_int64, [C.POINTER(EncConfig)]),
    "cbas_enc_create": (c_int, [C.POINTER(EncConfig), c_void_p, c_int64, c_int, C.POINTER(c_void_p)]),
    "cbas_enc_destroy": (None, [c_void_p]),
    "cbas_enc_forward_f32": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "cbas_enc_forward_u8": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int64, c_int64, c_int64,
                                    c_void_p, c_void_p, c_void_p]),
    "cbas_enc_submit_u8_host": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int64, c_int64, c_int64]),
    "cbas_enc_wait": (c_int, [c_void_p, c_int, c_void_p, c_void_p]),
    "cbas_enc_submit_u8": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int64, c_int64, c_int64, c_void_p,
                                   c_void_p, c_void_p]),
    "cbas_enc_wait_stream": (c_int, [c_void_p, c_int, c_void_p]),
    "cbas_enc_submit_u8_host_dev": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int64, c_int64, c_int64,
                                            c_void_p, c_void_p, c_void_p]),
    "cbas_enc_check_finite": (c_int, [c_void_p]),
    "cbas_enc_copy_stream": (c_void_p, [c_void_p]),
    "cbas_enc_get_config": (c_int, [c_void_p, C.POINTER(EncConfig)]),
    "cbas_head_get_config": (c_int, [c_void_p, C.POINTER(HeadConfigC)]),
    "cbas_fused_create": (c_int, [c_void_p, c_void_p, c_int64, c_float, c_int64, C.POINTER(c_void_p)]),
    "cbas_fused_destroy": (None, [c_void_p]),
    "cbas_fused_reset": (c_int, [c_void_p]),
    "cbas_fused_push_u8_host": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int64, c_int64, c_int64]),
    "cbas_fused_push_u8": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int64, c_int64, c_int64, c_void_p]),
    "cbas_fused_finish": (c_int, [c_void_p, c_void_p, c_void_p, C.POINTER(c_void_p), C.POINTER(c_void_p),
                                  C.POINTER(c_int64), c_void_p]),
    "cbas_fused_finish_async": (c_int, [c_void_p, c_void_p, c_void_p, C.POINTER(c_int64)]),
    "cbas_fused_wait": (c_int, [c_void_p]),
    "cbas_fused_stream_rows": (c_int, [c_void_p, c_void_p]),
    "cbas_fused_rows_ready": (c_int64, [c_void_p, c_int32]),
    "cbas_enc_set_lanes": (c_int, [c_void_p, c_int]),
    "cbas_enc_set_prune_last_layer": (c_int, [c_void_p, c_int]),
    "cbas_enc_profile": (c_int, [c_void_p, c_int]),
    "cbas_enc_profile_read": (c_int, [c_void_p, C.POINTER(C.c_double), C.POINTER(c_int64), C.POINTER(C.c_double),
                                      c_int]),
    "cbas_head_weights_count": (c_int64, [C.POINTER(HeadConfigC)]),
    "cbas_head_create": (c_int, [C.POINTER(HeadConfigC), c_void_p, c_int64, c_int, C.POINTER(c_void_p)]),
    "cbas_head_destroy": (None, [c_void_p]),
    "cbas_head_forward_windows": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_void_p]),
    "cbas_head_infer_f16": (c_int, [c_void_p, c_void_p, c_int64, c_float, c_void_p, c_void_p, c_void_p]),
    "cbas_head_infer_f16_range": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_int64, c_float, c_void_p, c_void_p,
                                          c_void_p]),
    "cbas_head_infer_f32_range": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_int64, c_float, c_void_p, c_void_p,
                                          c_void_p]),
    "cbas_head_train_create": (c_int, [C.POINTER(HeadConfigC), C.POINTER(TrainConfigC), c_void_p, c_int64, c_void_p, c_int,
                                       C.POINTER(c_void_p)]),
    "cbas_head_train_destroy": (None, [c_void_p]),
    "cbas_head_train_step": (c_int, [c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_void_p, c_void_p]),
    "cbas_head_train_read": (c_int, [c_void_p, c_int32, c_void_p, c_int64]),
    "cbas_head_train_last_outputs": (c_int, [c_void_p, c_void_p, c_void_p, c_int32]),
    "cbas_csv_format_f32": (c_int64, [c_void_p, c_int64, c_int32, c_void_p, c_int64]),
    "cbas_csv_write_f32": (c_int, [C.c_char_p, C.c_char_p, c_void_p, c_int64, c_int32, c_int32]),
    "cbas_mjpeg_decode": (c_int, [c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_void_p, c_int32, c_void_p]),
    "cbas_pick_channel_u8": (c_int, [c_void_p, c_int64, c_int32, c_int32, c_void_p, c_int32]),
    "cbas_last_error": (C.c_char_p, []),
    "cbas_abi_version": (c_int, []),
    "cbas_device_info": (c_int, [c_int, C.c_char_p, c_int, C.POINTER(c_int32), C.POINTER(c_int64)]),
}

# include/cbas_mi355x_debug.h: exported by libcbas_mi355x_debug.so only (CBAS_BUILD_DEBUG=1)
DEBUG_SIGNATURES = {
    "cbas_enc_debug_forward_u8": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int64, c_int64, c_int64,
                                          c_int, c_int]),
    "cbas_enc_debug_read": (c_int, [c_void_p, c_int, c_void_p, c_int64]),
    "cbas_enc_debug_option": (c_int, [c_void_p, C.c_char_p, c_int]),
    "cbas_debug_overlap": (c_int, [c_int, c_int, C.POINTER(c_float)]),
    "cbas_debug_gemm_bench": (c_int, [c_int, c_int, c_int, c_int, c_int, C.POINTER(c_float),
                                      C.POINTER(C.c_ulonglong)]),
    "cbas_debug_gemm_f8": (c_int, [c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                   c_void_p]),
    "cbas_debug_mfma_neighbor": (c_int, [c_int, c_void_p]),
    "cbas_debug_gemm_split_bench": (c_int, [c_int, c_int, c_int, c_int, c_int, c_int, C.POINTER(c_float)]),
    "cbas_head_debug_read": (c_int, [c_void_p, c_int, c_void_p, c_int64]),
    "cbas_debug_gemm_split_compare": (c_int, [c_int, c_int, c_int, c_int, c_int, C.POINTER(c_int64)]),
    "cbas_head_debug_expand_module": (c_int, [c_void_p, C.c_char_p, C.c_char_p]),
    "cbas_head_debug_expand_repeat": (c_int, [c_void_p, c_int, c_int]),
    "cbas_head_debug_expand_stats": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int]),
    "cbas_debug_build": (c_int, []),
}

ENC_SLOTS = 3
EXPECTED_ABI = 10         # CBAS_ABI_VERSION of include/cbas_mi355x.h these ctypes structures mirror
PROF_CATS = ["patch_gemm", "layernorm", "qkv_gemm", "attention", "oproj_gemm", "up_gemm", "down_gemm", "other"]


def library_path() -> str:
    return _build.lib_path()


def is_debug() -> bool:
    """True when the loaded (or to-be-loaded) library is the debug build."""
    return _build.debug_selected()


def require_debug(what: str) -> None:
    if not is_debug():
        raise RuntimeError(f"{what} is a bring-up / test entry point (include/cbas_mi355x_debug.h): set CBAS_BUILD_DEBUG=1 "
                           "before the first cbas_amd call to load libcbas_mi355x_debug.so (python -m cbas_amd.build --debug)")


def load(build_if_missing: bool = True):
    """Load (building first if needed) the native library; raises if it cannot be had."""
    global _lib
    with _lock:
        if _lib is not None:
            return _lib
        # torch ships its own libamdhip64; import it FIRST so this library binds to the same HIP
        # runtime (two runtimes in one process do not see the GPU: "no ROCm-capable device")
        import torch  # noqa: F401
        path = _build.lib_path()
        debug = _build.debug_selected()
        if not os.path.exists(path):
            if not build_if_missing:
                raise RuntimeError(f"{path} is missing; run `python -m cbas_amd.build{' --debug' if debug else ''}`")
            _build.build_library()
        elif _build._stale(debug):
            # sources newer than the binary: rebuild when asked to (CBAS_AUTOBUILD=1), otherwise say so -
            # a silent stale binary is how an edited kernel "does nothing"
            if os.environ.get("CBAS_AUTOBUILD") == "1":
                _build.build_library()
            else:
                import sys
                print(f"cbas_amd: {path} is older than its sources; run `python -m cbas_amd.build` "
                      "(or set CBAS_AUTOBUILD=1)", file=sys.stderr)
        lib = C.CDLL(path)
        for name, (res, args) in list(SIGNATURES.items()) + (list(DEBUG_SIGNATURES.items()) if debug else []):
            fn = getattr(lib, name)      # AttributeError if the .so lacks a declared symbol
            fn.restype = res
            fn.argtypes = args
        abi = lib.cbas_abi_version()
        if abi != EXPECTED_ABI:          # struct layouts below would not match: fail before any config is passed
            raise RuntimeError(f"{path} implements ABI v{abi}, these bindings are for v{EXPECTED_ABI}; rebuild with "
                               "`python -m cbas_amd.build --force`")
        _lib = lib
        return lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = load().cbas_last_error()
        raise RuntimeError(f"{what} failed (code {rc}): {msg.decode(errors='replace') if msg else '?'}")


def device_info(device_id: int = 0):
    lib = load()
    buf = C.create_string_buffer(64)
    ncu, hbm = c_int32(0), c_int64(0)
    check(lib.cbas_device_info(device_id, buf, 64, C.byref(ncu), C.byref(hbm)), "cbas_device_info")
    return buf.value.decode(), int(ncu.value), int(hbm.value)
