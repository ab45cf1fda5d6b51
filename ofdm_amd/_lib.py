"""ctypes loader for libofdm_hip.so.  Fails loudly: there is no CPU fallback in the product path."""
from __future__ import annotations

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libofdm_hip.so")
_lib = None
_profile = False


def use_profile_build():
    """tools/ only: load libofdm_hip_profile.so (the build with the kernels' ablation exits and section timers, reachable
    through ofdm_set_tuning "debug_*") instead of the product library.  Must be called before the first load()."""
    global _profile
    if _lib is not None and not _profile:
        raise RuntimeError("use_profile_build() must precede the first load()")
    _profile = True

i32, i64, u8p, vp = C.c_int32, C.c_int64, C.c_void_p, C.c_void_p


class Params(C.Structure):
    """ofdm_params (include/ofdm_hip.h)."""

    _fields_ = [
        ("n_fft", C.c_int32), ("cp_len", C.c_int32), ("modulation", C.c_int32), ("guard_bands", C.c_int32),
        ("ecc", C.c_int32), ("sync_window_reps", C.c_int32), ("sync_backoff", C.c_int32), ("cfo_mode", C.c_int32),
        ("sync_threshold", C.c_float), ("sync_mode", C.c_int32), ("rx_path", C.c_int32), ("reserved", C.c_int32 * 5),
    ]


# name -> (restype, argtypes); every symbol declared in include/ofdm_hip.h
SIGNATURES = {
    "ofdm_abi_version": (C.c_int, []),
    "ofdm_strerror": (C.c_char_p, [C.c_int]),
    "ofdm_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "ofdm_default_params": (C.c_int, [C.POINTER(Params)]),
    "ofdm_default_pilots": (C.c_int, [i32, i32, vp, vp]),
    "ofdm_stdrng_pilots": (C.c_int, [i32, i32, vp, vp]),
    "ofdm_chacha_block": (C.c_int, [vp, vp, i32, vp]),
    "ofdm_tx_symbols_batch": (C.c_int, [vp, vp, i64, vp, i64]),
    "ofdm_rs255_encoded_len": (C.c_int64, [i64]),
    "ofdm_rs255_decoded_len": (C.c_int64, [i64]),
    "ofdm_rs255_encode": (C.c_int, [vp, i64, vp]),
    "ofdm_rs255_decode": (C.c_int, [vp, i64, vp, vp]),
    "ofdm_create": (C.c_int, [C.POINTER(Params), vp, vp, C.c_int, vp, C.POINTER(vp)]),
    "ofdm_destroy": (C.c_int, [vp]),
    "ofdm_set_stream": (C.c_int, [vp, vp]),
    "ofdm_use_own_stream": (C.c_int, [vp]),
    "ofdm_synchronize": (C.c_int, [vp]),
    "ofdm_last_hip_error": (C.c_int, [vp]),
    "ofdm_last_dispatch": (C.c_int, [vp, C.c_char_p, C.c_size_t]),
    "ofdm_set_tuning": (C.c_int, [vp, C.c_char_p, i64]),
    "ofdm_get_tuning": (C.c_int, [vp, C.c_char_p, C.POINTER(i64)]),
    "ofdm_dev_alloc": (C.c_int, [vp, C.c_size_t, C.POINTER(vp)]),
    "ofdm_dev_free": (C.c_int, [vp, vp]),
    "ofdm_memcpy_h2d": (C.c_int, [vp, vp, vp, C.c_size_t]),
    "ofdm_memcpy_d2h": (C.c_int, [vp, vp, vp, C.c_size_t]),
    "ofdm_memset": (C.c_int, [vp, vp, C.c_int, C.c_size_t]),
    "ofdm_symbol_len": (C.c_int, [vp]),
    "ofdm_data_carriers": (C.c_int, [vp]),
    "ofdm_bytes_per_symbol": (C.c_int, [vp]),
    "ofdm_coded_len": (i64, [vp, i64]),
    "ofdm_data_symbols": (i64, [vp, i64]),
    "ofdm_frame_samples": (i64, [vp, i64]),
    "ofdm_fft_batch": (C.c_int, [vp, vp, vp, i64, C.c_int]),
    "ofdm_ifft_cp_batch": (C.c_int, [vp, vp, vp, i64]),
    "ofdm_unprefix_batch": (C.c_int, [vp, vp, vp, i64]),
    "ofdm_qam_map_batch": (C.c_int, [vp, vp, i64, vp]),
    "ofdm_qam_demap_batch": (C.c_int, [vp, vp, i64, vp, vp]),
    "ofdm_encode_block_batch": (C.c_int, [vp, vp, vp, i64]),
    "ofdm_normalize_batch": (C.c_int, [vp, vp, i64, i64, i64]),
    "ofdm_hamming74_encode": (C.c_int, [vp, vp, i64, vp]),
    "ofdm_hamming74_decode": (C.c_int, [vp, vp, i64, vp, vp]),
    "ofdm_sc_correlate_batch": (C.c_int, [vp, vp, i64, i64, i64, i64, vp, vp, vp]),
    "ofdm_frequency_correction_batch": (C.c_int, [vp, vp, i64, i64, i64, vp]),
    "ofdm_cfo_rotate_batch": (C.c_int, [vp, vp, i64, i64, i64, vp, vp]),
    "ofdm_estimate_channel_batch": (C.c_int, [vp, vp, i64, i64, i64, vp, vp, vp]),
    "ofdm_rx_demod_batch": (C.c_int, [vp, vp, i64, i64, i64, i32, i32, vp, vp, vp, i64, vp, i64, vp]),
    "ofdm_tx_encode_batch": (C.c_int, [vp, vp, i64, i64, vp, i32, vp, i64]),
    "ofdm_rx_decode_batch": (C.c_int, [vp, vp, i64, i64, i64, i64, i32, vp, i64, vp, vp, vp, vp, vp]),
    "ofdm_xcorr_batch": (C.c_int, [vp, vp, i64, i64, i64, vp, i32, vp, vp, vp, i64]),
    "ofdm_channel_batch": (C.c_int, [vp, vp, i64, i64, i64, C.c_double, i32, C.c_uint64, vp, vp, vp, i64, i64, vp]),
    "ofdm_channel_taps": (C.c_int, [vp]),
    "ofdm_host_alloc": (C.c_int, [C.c_size_t, C.POINTER(vp)]),
    "ofdm_host_free": (C.c_int, [vp]),
    "ofdm_host_register": (C.c_int, [vp, C.c_size_t]),
    "ofdm_host_unregister": (C.c_int, [vp]),
    "ofdm_host_is_pinned": (C.c_int, [vp, C.c_size_t]),
    "ofdm_rx_decode_host": (C.c_int, [vp, vp, i64, i64, i64, i64, i32, vp, i64, vp, vp, vp, vp, vp, i64]),
    "ofdm_rx_demod_host": (C.c_int, [vp, vp, i64, i64, i64, i32, i32, vp, i64, i64]),
    "ofdm_tx_encode_host": (C.c_int, [vp, vp, i64, i64, vp, i32, vp, i64, i64]),
    "ofdm_sc_correlate_long": (C.c_int, [vp, vp, i64, i64, i64, i64, C.POINTER(i64), C.POINTER(C.c_double), C.POINTER(C.c_float)]),
    "ofdm_rx_decode_long": (C.c_int, [vp, vp, i64, i64, i64, i64, i32, vp, i64, C.POINTER(i32), C.POINTER(i32), C.POINTER(i64),
                                      C.POINTER(C.c_double), C.POINTER(C.c_float)]),
    "ofdm_rx_decode_long_host": (C.c_int, [vp, vp, i64, i32, vp, i64, C.POINTER(i32), C.POINTER(i32), C.POINTER(i64),
                                           C.POINTER(C.c_double), C.POINTER(C.c_float)]),
    "ofdm_hbm_read_probe": (C.c_int, [vp, vp, i64, i32]),
    "ofdm_timer_start": (C.c_int, [vp]),
    "ofdm_timer_stop_ms": (C.c_int, [vp, C.POINTER(C.c_float)]),
}


def load() -> C.CDLL:
    """Load libofdm_hip.so, building it with hipcc first if it is missing.  Raises if that is impossible."""
    global _lib
    if _lib is not None:
        return _lib
    path = LIB_PATH
    if _profile:
        from . import build as _build

        path = _build.build(profile=True)
    elif not os.path.exists(LIB_PATH):
        from . import build as _build

        _build.build()
    # libofdm_hip.so needs libamdhip64.so.7.  torch ships its own copy under the same soname: import torch
    # first so that ONE HIP runtime is shared by torch (device memory, streams, RCCL) and by this library.
    import torch  # noqa: F401

    try:
        lib = C.CDLL(path, mode=C.RTLD_GLOBAL)
    except OSError as e:  # no silent fallback
        raise RuntimeError(f"cannot load {path}: {e}") from e
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the library lacks a declared symbol
        fn.restype = res
        fn.argtypes = args
    if lib.ofdm_abi_version() != 1:
        raise RuntimeError("libofdm_hip.so ABI version mismatch")
    _lib = lib
    return lib
