"""ctypes view of the CPU oracle (oracle/ofdm_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg as the checker.  The product package (ofdm_amd) never imports this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

BPSK, QPSK, QAM16, QAM64, QAM256 = 1, 2, 4, 6, 8


class RxInfo(C.Structure):
    _fields_ = [
        ("status", C.c_int),
        ("offset", C.c_long),
        ("f_delta", C.c_double),
        ("metric", C.c_double),
        ("n_bytes", C.c_size_t),
        ("n_symbols", C.c_size_t),
    ]


def build(force: bool = False) -> str:
    so = os.path.join(_HERE, "liboracle.so")
    src = [os.path.join(_HERE, f) for f in ("ofdm_oracle.c", "ofdm_oracle.h")]
    stale = (not os.path.exists(so)) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in src)
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "liboracle.so"], stdout=subprocess.DEVNULL)
    return so


def lib() -> C.CDLL:
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(build())
        L = _LIB
        L.orc_uniform_pm1.restype = C.c_double
        L.orc_uniform_01.restype = C.c_double
        L.orc_splitmix64.restype = C.c_uint64
        L.orc_angle.restype = C.c_double
        L.orc_frequency_correction.restype = C.c_double
        for name in ("orc_modulate_count", "orc_modulate", "orc_frame_len", "orc_encode", "orc_decode_block",
                     "orc_demodulate", "orc_hamming74_encoded_len", "orc_hamming74_encode",
                     "orc_hamming74_decode", "orc_rx_demod", "orc_channel"):
            getattr(L, name).restype = C.c_size_t
        for name in ("orc_decode_ref", "orc_decode_sc", "orc_decode_given"):
            getattr(L, name).restype = RxInfo
        L.orc_bools_to_u8.restype = C.c_uint8
    return _LIB


class _C64(C.Structure):
    _fields_ = [("re", C.c_double), ("im", C.c_double)]


def _c(a: np.ndarray):
    """pointer to a contiguous complex128 array (or None)."""
    if a is None:
        return None
    assert a.dtype == np.complex128 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(C.c_void_p)


def _u8(a: np.ndarray):
    assert a.dtype == np.uint8 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(C.c_void_p)


def set_fft_cache(on: bool):
    """False (default): libm twiddles on every transform (the reference re-plans per call, signals/mod.rs:41-58);
    True: cached per-thread tables.  Same results bit for bit."""
    lib().orc_set_fft_cache(C.c_int(int(on)))


def cx(x) -> np.ndarray:
    return np.ascontiguousarray(np.asarray(x, dtype=np.complex128))


# ---------------------------------------------------------------- primitives
def fft(x, inverse=False):
    a = cx(x).copy()
    lib().orc_fft(_c(a), C.c_int(a.size), C.c_int(int(inverse)))
    return a


def fft_shift(x):
    a = cx(x).copy()
    lib().orc_fft_shift(_c(a), C.c_int(a.size))
    return a


def ifft_shift(x):
    a = cx(x).copy()
    lib().orc_ifft_shift(_c(a), C.c_int(a.size))
    return a


def xcorr_fft(a, b):
    a, b = cx(a), cx(b)
    out = np.zeros(2 * a.size - 1, np.complex128)
    idx = lib().orc_xcorr_fft(_c(a), C.c_int(a.size), _c(b), C.c_int(b.size), _c(out))
    return int(idx), out


def convolve(a, b):
    a, b = cx(a), cx(b)
    out = np.zeros(a.size + b.size - 1, np.complex128)
    lib().orc_convolve(_c(a), C.c_int(a.size), _c(b), C.c_int(b.size), _c(out))
    return out


def mean(x):
    L = lib()
    L.orc_mean.restype = _C64
    a = cx(x)
    r = L.orc_mean(_c(a), C.c_int(a.size))
    return complex(r.re, r.im)


def variance(x):
    L = lib()
    L.orc_variance.restype = _C64
    a = cx(x)
    r = L.orc_variance(_c(a), C.c_int(a.size))
    return complex(r.re, r.im)


def angle(z: complex) -> float:
    return float(lib().orc_angle(_C64(z.real, z.imag)))


def to_bools(b: int):
    out = (C.c_uint8 * 8)()
    lib().orc_to_bools(C.c_uint8(b), out)
    return [bool(v) for v in out]


def bools_to_u8(bools) -> int:
    arr = (C.c_uint8 * 8)(*[1 if b else 0 for b in bools])
    return int(lib().orc_bools_to_u8(arr))


def analysis(left: bytes, right: bytes):
    l = np.frombuffer(bytes(left), np.uint8)
    r = np.frombuffer(bytes(right), np.uint8)
    assert l.size == r.size
    ne, nb, rate = C.c_uint32(), C.c_uint32(), C.c_double()
    lib().orc_analysis(_u8(np.ascontiguousarray(l)), _u8(np.ascontiguousarray(r)), C.c_size_t(l.size),
                       C.byref(ne), C.byref(nb), C.byref(rate))
    return ne.value, nb.value, rate.value


def sig_to_fc32(x) -> np.ndarray:
    a = cx(x)
    out = np.zeros(2 * a.size, np.float32)
    lib().orc_sig_to_fc32(_c(a), C.c_size_t(a.size), out.ctypes.data_as(C.c_void_p))
    return out


def fc32_to_sig(f) -> np.ndarray:
    f = np.ascontiguousarray(f, dtype=np.float32).reshape(-1)
    out = np.zeros(f.size // 2, np.complex128)
    lib().orc_fc32_to_sig(f.ctypes.data_as(C.c_void_p), C.c_size_t(out.size), _c(out))
    return out


# ---------------------------------------------------------------- carriers / pilots
def carrier_class(bin_, n_fft=64, guard=True) -> int:
    return int(lib().orc_carrier_class(C.c_int(bin_), C.c_int(n_fft), C.c_int(int(guard))))


def data_carriers(n_fft=64, guard=True) -> int:
    return int(lib().orc_data_carriers(C.c_int(n_fft), C.c_int(int(guard))))


def locking_signal(length=80):
    out = np.zeros(length, np.complex128)
    lib().orc_locking_signal(C.c_int(length), _c(out))
    return out


def default_preamble(length=80):
    out = np.zeros(length, np.complex128)
    lib().orc_default_preamble(C.c_int(length), _c(out))
    return out


def default_training(length=64):
    out = np.zeros(length, np.complex128)
    lib().orc_default_training(C.c_int(length), _c(out))
    return out


def stdrng_preamble(length=80):
    """transmitter.rs:75-84 with rand 0.8 StdRng (ChaCha12) restated; unverified against a running `rand`."""
    out = np.zeros(length, np.complex128)
    lib().orc_stdrng_preamble(C.c_int(length), _c(out))
    return out


def stdrng_training(length=64):
    out = np.zeros(length, np.complex128)
    lib().orc_stdrng_training(C.c_int(length), _c(out))
    return out


def rs_encode(msg: bytes, nsym: int) -> bytes:
    m = np.frombuffer(bytes(msg), np.uint8).copy()
    out = np.zeros(m.size + nsym, np.uint8)
    assert lib().orc_rs_encode(_u8(m), C.c_int(m.size), C.c_int(nsym), _u8(out)) == 0
    return bytes(out)


def rs_correct(cw: bytes, nsym: int):
    c = np.frombuffer(bytes(cw), np.uint8).copy()
    n = lib().orc_rs_correct(_u8(c), C.c_int(c.size), C.c_int(nsym))
    return (bytes(c), n)


def create_transmission_bytes(data: bytes) -> bytes:
    d = np.frombuffer(bytes(data), np.uint8).copy()
    out = np.zeros(255 * (d.size // 223 + 1), np.uint8)
    lib().orc_create_transmission_bytes.restype = C.c_long
    n = lib().orc_create_transmission_bytes(_u8(d) if d.size else None, C.c_long(d.size), _u8(out))
    return bytes(out[:n])


def decipher_transmission_bytes(code: bytes):
    c = np.frombuffer(bytes(code), np.uint8).copy()
    out = np.zeros(223 * (c.size // 255 + 1), np.uint8)
    lib().orc_decipher_transmission_bytes.restype = C.c_long
    n = lib().orc_decipher_transmission_bytes(_u8(c) if c.size else None, C.c_long(c.size), _u8(out))
    return None if n < 0 else bytes(out[:n])


def chacha_block(state16, rounds):
    st = np.ascontiguousarray(state16, dtype=np.uint32)
    out = np.zeros(16, np.uint32)
    lib().orc_chacha_keystream_block(st.ctypes.data_as(C.c_void_p), C.c_int(rounds), out.ctypes.data_as(C.c_void_p))
    return out


# ---------------------------------------------------------------- TX
def modulate(data: bytes, modulation: int) -> np.ndarray:
    d = np.frombuffer(bytes(data), np.uint8).copy()
    n = lib().orc_modulate_count(C.c_size_t(d.size), C.c_int(modulation))
    out = np.zeros(max(n, 1), np.complex128)
    lib().orc_modulate(_u8(d) if d.size else None, C.c_size_t(d.size), C.c_int(modulation), _c(out))
    return out[:n]


def encode_block(stream, n_fft=64, guard=False):
    s = cx(stream)
    out = np.zeros(n_fft, np.complex128)
    used = C.c_size_t()
    lib().orc_encode_block(_c(s) if s.size else None, C.c_size_t(s.size), C.byref(used), C.c_int(n_fft),
                           C.c_int(int(guard)), _c(out))
    return out, used.value


def prefix_block(freq, cp=None):
    f = cx(freq)
    cp = f.size // 4 if cp is None else cp
    out = np.zeros(f.size + cp, np.complex128)
    lib().orc_prefix_block(_c(f), C.c_int(f.size), C.c_int(cp), _c(out))
    return out


def normalize(x):
    a = cx(x).copy()
    with np.errstate(all="ignore"):
        lib().orc_normalize(_c(a), C.c_size_t(a.size))
    return a


def frame_len(payload_bytes, n_fft=64, guard=False, modulation=BPSK) -> int:
    return int(lib().orc_frame_len(C.c_size_t(payload_bytes), C.c_int(n_fft), C.c_int(n_fft // 4),
                                   C.c_int(int(guard)), C.c_int(modulation)))


def encode(data: bytes, guard=False, modulation=BPSK, n_fft=64, preamble=None, training=None) -> np.ndarray:
    cp = n_fft // 4
    d = np.frombuffer(bytes(data), np.uint8).copy()
    pre = default_preamble(n_fft + cp) if preamble is None else cx(preamble)
    trn = default_training(n_fft) if training is None else cx(training)
    out = np.zeros(frame_len(d.size, n_fft, guard, modulation), np.complex128)
    n = lib().orc_encode(_u8(d) if d.size else None, C.c_size_t(d.size), C.c_int(n_fft), C.c_int(cp),
                         C.c_int(int(guard)), C.c_int(modulation), _c(pre), _c(trn), _c(out))
    assert n == out.size
    return out


# ---------------------------------------------------------------- RX
def unprefix_block(block, n_fft=64):
    b = cx(block)
    cp = b.size - n_fft
    out = np.zeros(n_fft, np.complex128)
    lib().orc_unprefix_block(_c(b), C.c_int(n_fft), C.c_int(cp), _c(out))
    return out


def decode_block(freq, guard=False):
    f = cx(freq)
    out = np.zeros(f.size, np.complex128)
    n = lib().orc_decode_block(_c(f), C.c_int(f.size), C.c_int(int(guard)), _c(out))
    return out[:n]


def demodulate(sym, modulation) -> bytes:
    s = cx(sym)
    out = np.zeros(s.size * modulation // 8 + 1, np.uint8)
    n = lib().orc_demodulate(_c(s), C.c_size_t(s.size), C.c_int(modulation), _u8(out))
    return bytes(out[:n])


def demap_indices(sym, modulation) -> np.ndarray:
    s = cx(sym)
    out = np.zeros(s.size, np.uint8)
    lib().orc_demap_indices(_c(s), C.c_size_t(s.size), C.c_int(modulation), _u8(out))
    return out


def estimate_channel(blocks, training, n_fft=64):
    b = cx(blocks).reshape(-1)
    cp = n_fft // 4
    assert b.size == 5 * (n_fft + cp)
    hk = np.zeros(n_fft, np.complex128)
    lib().orc_estimate_channel(_c(b), C.c_int(n_fft), C.c_int(cp), _c(cx(training)), _c(hk))
    return hk


def frequency_correction(left, right) -> float:
    l, r = cx(left), cx(right)
    return float(lib().orc_frequency_correction(_c(l), _c(r), C.c_int(l.size)))


def cfo_rotate(x, f_delta, first_index=0):
    a = cx(x).copy()
    lib().orc_cfo_rotate(_c(a), C.c_size_t(a.size), C.c_double(f_delta), C.c_size_t(first_index))
    return a


def sc_sync(r, L=80, window_reps=3, n_lags=0, threshold=0.5):
    a = cx(r)
    p = _C64()
    metric, fd = C.c_double(), C.c_double()
    d = lib().orc_sc_sync(_c(a), C.c_size_t(a.size), C.c_int(L), C.c_int(window_reps), C.c_long(n_lags),
                          C.c_double(threshold), C.byref(p), C.byref(metric), C.byref(fd))
    return int(d), complex(p.re, p.im), metric.value, fd.value


def sc_metric(r, L=80, window_reps=3, n_lags=0):
    a = cx(r)
    valid = a.size - (window_reps + 1) * L + 1
    n = valid if n_lags <= 0 or n_lags > valid else n_lags
    m = np.zeros(max(n, 0), np.float64)
    p = np.zeros(max(n, 0), np.complex128)
    if n > 0:
        lib().orc_sc_metric(_c(a), C.c_size_t(a.size), C.c_int(L), C.c_int(window_reps), C.c_long(n_lags),
                            m.ctypes.data_as(C.c_void_p), _c(p))
    return m, p


def hamming74_encode(data: bytes) -> bytes:
    d = np.frombuffer(bytes(data), np.uint8).copy()
    out = np.zeros(lib().orc_hamming74_encoded_len(C.c_size_t(d.size)) + 1, np.uint8)
    n = lib().orc_hamming74_encode(_u8(d) if d.size else None, C.c_size_t(d.size), _u8(out))
    return bytes(out[:n])


def hamming74_decode(code: bytes):
    c = np.frombuffer(bytes(code), np.uint8).copy()
    out = np.zeros(c.size // 7 * 4 + 1, np.uint8)
    fixed = C.c_uint32()
    n = lib().orc_hamming74_decode(_u8(c) if c.size else None, C.c_size_t(c.size), _u8(out), C.byref(fixed))
    return bytes(out[:n]), fixed.value


def _rx_buffers(n_samples, n_fft, modulation, want_soft):
    cap = n_samples  # generous: never more bytes than samples
    out = np.zeros(cap + 16, np.uint8)
    soft = np.zeros(n_samples, np.complex128) if want_soft else None
    return out, soft


def _rx_result(info, out, soft, n_fft, guard):
    res = {
        "status": info.status, "offset": info.offset, "f_delta": info.f_delta, "metric": info.metric,
        "bytes": bytes(out[: info.n_bytes]) if info.status == 0 else b"", "n_symbols": info.n_symbols,
    }
    if soft is not None:
        res["soft"] = soft[: info.n_symbols * data_carriers(n_fft, guard)].copy()
    return res


def decode_ref(samples, guard=False, modulation=BPSK, n_fft=64, training=None, want_soft=False):
    s = cx(samples)
    trn = default_training(n_fft) if training is None else cx(training)
    out, soft = _rx_buffers(s.size, n_fft, modulation, want_soft)
    info = lib().orc_decode_ref(_c(s), C.c_size_t(s.size), C.c_int(n_fft), C.c_int(n_fft // 4), C.c_int(int(guard)),
                                C.c_int(modulation), _c(trn), _u8(out), C.c_size_t(out.size), _c(soft),
                                C.c_size_t(0 if soft is None else soft.size))
    return _rx_result(info, out, soft, n_fft, guard)


def decode_sc(samples, guard=False, modulation=BPSK, n_fft=64, training=None, window_reps=3, sync_lags=0,
              threshold=0.5, backoff=4, cfo_abs=False, max_symbols=0, want_soft=False, cfo_off=False):
    s = cx(samples)
    trn = default_training(n_fft) if training is None else cx(training)
    out, soft = _rx_buffers(s.size, n_fft, modulation, want_soft)
    info = lib().orc_decode_sc(_c(s), C.c_size_t(s.size), C.c_int(n_fft), C.c_int(n_fft // 4), C.c_int(int(guard)),
                               C.c_int(modulation), _c(trn), C.c_int(window_reps), C.c_long(sync_lags),
                               C.c_double(threshold), C.c_int(backoff), C.c_int(2 if cfo_off else int(cfo_abs)), C.c_int(max_symbols), _u8(out),
                               C.c_size_t(out.size), _c(soft), C.c_size_t(0 if soft is None else soft.size))
    return _rx_result(info, out, soft, n_fft, guard)


def decode_given(samples, offset, f_delta, guard=False, modulation=BPSK, n_fft=64, training=None, max_symbols=0,
                 want_soft=False):
    s = cx(samples)
    trn = default_training(n_fft) if training is None else cx(training)
    out, soft = _rx_buffers(s.size, n_fft, modulation, want_soft)
    info = lib().orc_decode_given(_c(s), C.c_size_t(s.size), C.c_long(offset), C.c_double(f_delta), C.c_int(n_fft),
                                  C.c_int(n_fft // 4), C.c_int(int(guard)), C.c_int(modulation), _c(trn),
                                  C.c_int(max_symbols), _u8(out), C.c_size_t(out.size), _c(soft),
                                  C.c_size_t(0 if soft is None else soft.size))
    return _rx_result(info, out, soft, n_fft, guard)


def rx_demod(samples, n_fft=64, guard=True, modulation=QAM64, hk=None, want_soft=False):
    s = cx(samples)
    S = n_fft + n_fft // 4
    nsym = s.size // S
    nd = data_carriers(n_fft, guard)
    out = np.zeros(nsym * nd * modulation // 8 + 1, np.uint8)
    soft = np.zeros(nsym * nd, np.complex128) if want_soft else None
    n = lib().orc_rx_demod(_c(s), C.c_size_t(nsym), C.c_int(n_fft), C.c_int(n_fft // 4), C.c_int(int(guard)),
                           C.c_int(modulation), _c(None if hk is None else cx(hk)), _u8(out), _c(soft))
    return (bytes(out[:n]), soft) if want_soft else bytes(out[:n])


def channel_taps() -> np.ndarray:
    return np.ctypeslib.as_array((C.c_double * 64).in_dll(lib(), "ORC_CHANNEL")).copy()


def channel(tx, snr_db=30.0, timing_error=False, seed=1):
    t = cx(tx)
    out = np.zeros(t.size + 63, np.complex128)
    fd = C.c_double()
    lib().orc_channel(_c(t), C.c_size_t(t.size), C.c_double(snr_db), C.c_int(int(timing_error)), C.c_uint64(seed),
                      _c(out), C.byref(fd))
    return out, fd.value
