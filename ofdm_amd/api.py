"""Host-side mirror of the reference's TX/RX function surface over the C ABI (include/ofdm_hip.h).

The reference's hot path is a set of free functions re-exported from its crate root (src/lib.rs:8-21):
`encode`, `decode`, `modulate`, `demodulate`, `encode_block`, `prefix_block`, `unprefix_block`,
`estimate_channel`, `frequency_correction`, `normalize`, `locking_signal`, `preamble`, `training_signals`.
This module keeps those names, argument meanings, defaults and error behaviour, batch-first, with the DSP
bodies running as HIP kernels on one MI355X.  torch is used for device memory and streams only: every
computation goes through libofdm_hip.so, and nothing here falls back to the CPU.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import numpy as np
import torch

from . import _lib

BPSK, QPSK, QAM16, QAM64, QAM256 = 1, 2, 4, 6, 8  # ModulationScheme (src/transmitter.rs:98-104) as bits/point
ECC_NONE, ECC_HAMMING74 = 0, 1
CFO_OFF, CFO_SIGNED, CFO_ABS = 0, 1, 2
FRAME_OK, FRAME_SHORT, FRAME_NOSYNC, FRAME_BADTIMING, FRAME_HEADER = 0, -1, -2, -3, -4
SYNC_SCHMIDL_COX, SYNC_REFERENCE = 0, 1
RX_AUTO, RX_STAGED = 0, 1
DEFAULT_TUNING: dict = {}  # merged under every Context's `tuning=` (tools/tune_env.py fills it; empty in tests, bench and smoke)


class OfdmError(RuntimeError):
    pass


class DecodeError(OfdmError):
    """`Err(anyhow!(..))` of decode (src/receiver.rs:27-29)."""


def _check(lib, rc: int, what: str, ctx=None):
    if rc != 0:
        extra = f" (hipError {lib.ofdm_last_hip_error(ctx)})" if ctx is not None and rc == -4 else ""
        raise OfdmError(f"{what}: {lib.ofdm_strerror(rc).decode()}{extra}")


def default_pilots(n_fft: int = 64):
    """(preamble[n_fft+cp], training[n_fft]) complex128 -- SplitMix64 stand-ins for the StdRng tables of
    src/transmitter.rs:75-96 (same seeds 100 / 50, same draw order and scaling)."""
    lib = _lib.load()
    cp = n_fft // 4
    pre = np.zeros(n_fft + cp, np.complex128)
    trn = np.zeros(n_fft, np.complex128)
    _check(lib, lib.ofdm_default_pilots(n_fft, cp, pre.ctypes.data, trn.ctypes.data), "ofdm_default_pilots")
    return pre, trn


def stdrng_pilots(n_fft: int = 64):
    """(preamble[n_fft+cp], training[n_fft]) as the reference draws them: rand 0.8 StdRng (ChaCha12), seeds 100 / 50
    (src/transmitter.rs:75-96).  Restated from the published algorithm; unverified against a running `rand`."""
    lib = _lib.load()
    cp = n_fft // 4
    pre = np.zeros(n_fft + cp, np.complex128)
    trn = np.zeros(n_fft, np.complex128)
    _check(lib, lib.ofdm_stdrng_pilots(n_fft, cp, pre.ctypes.data, trn.ctypes.data), "ofdm_stdrng_pilots")
    return pre, trn


def create_transmission_bytes(data: bytes) -> bytes:
    """utils::create_transmission_bytes (src/utils.rs:97-136): outer RS(255,223) blocks, host side."""
    lib = _lib.load()
    src = np.frombuffer(bytes(data), np.uint8).copy()
    out = np.zeros(lib.ofdm_rs255_encoded_len(src.size), np.uint8)
    _check(lib, lib.ofdm_rs255_encode(src.ctypes.data if src.size else None, src.size, out.ctypes.data), "ofdm_rs255_encode")
    return bytes(out)


def decipher_transmission_bytes(code: bytes) -> Optional[bytes]:
    """utils::decipher_transmission_bytes (src/utils.rs:150-180): None where the reference returns None
    (a block with more than 16 byte errors)."""
    lib = _lib.load()
    src = np.frombuffer(bytes(code), np.uint8).copy()
    out = np.zeros(lib.ofdm_rs255_decoded_len(src.size), np.uint8)
    rc = lib.ofdm_rs255_decode(src.ctypes.data if src.size else None, src.size, out.ctypes.data, None)
    if rc == -6:
        return None
    _check(lib, rc, "ofdm_rs255_decode")
    return bytes(out)


def sig_to_bytes(sig) -> bytes:
    """utils::sig_to_bytes (src/utils.rs:228-236): Complex64 samples -> native-endian f32 (re, im) pairs, the `fc32` format of
    UHD's `--type float` tools (data/transmit.sh:1) and the layout every kernel of this library reads and writes."""
    return np.ascontiguousarray(np.asarray(sig), dtype=np.complex64).tobytes()


def bytes_to_sig(buf: bytes) -> np.ndarray:
    """utils::bytes_to_sig (src/utils.rs:238-254): fc32 bytes -> Complex64 samples (a trailing partial sample is dropped, as
    `as_chunks` drops it)."""
    n = len(buf) // 8
    return np.frombuffer(buf, dtype=np.complex64, count=n).astype(np.complex128)


def locking_signal(length: int = 80) -> np.ndarray:
    """src/transmitter.rs:60-72 (host-side constant; the library builds the same table into its frame header)."""
    v = 0.5 * (np.arange(length) / (2.0 * length) + 0.5)
    return np.fft.fftshift(v).astype(np.complex128) if length % 2 == 0 else np.roll(v, -((length + 1) // 2)) + 0j


def preamble(length: int = 80) -> np.ndarray:
    return default_pilots(length * 4 // 5)[0]


def training_signals(length: int = 64) -> np.ndarray:
    return default_pilots(length)[1]


def pinned_empty(shape, dtype) -> np.ndarray:
    """A numpy array in page-locked host memory (ofdm_host_alloc): what the host-buffer entry points DMA in place.  The memory
    is returned to the driver when the array (and every view of it) is gone."""
    import weakref

    lib = _lib.load()
    dt = np.dtype(dtype)
    n = int(np.prod(shape)) * dt.itemsize
    ptr = C.c_void_p()
    _check(lib, lib.ofdm_host_alloc(max(n, 1), C.byref(ptr)), "ofdm_host_alloc")
    buf = (C.c_char * max(n, 1)).from_address(ptr.value)
    arr = np.frombuffer(buf, dtype=np.uint8, count=n).view(dt).reshape(shape)
    weakref.finalize(buf, lib.ofdm_host_free, C.c_void_p(ptr.value))
    return arr


def _host(a: Optional[np.ndarray]):
    return None if a is None else C.c_void_p(a.ctypes.data)


def _dev(t: Optional[torch.Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


class Context:
    """One (GPU, stream) context = `ofdm_ctx`.  Tensors passed in must live on this context's device."""

    def __init__(self, n_fft: int = 64, modulation: int = BPSK, guard_bands: bool = False, ecc: int = ECC_NONE,
                 device: int = 0, preamble: Optional[np.ndarray] = None, training: Optional[np.ndarray] = None,
                 sync_window_reps: int = 3, sync_backoff: int = 4, cfo_mode: int = CFO_SIGNED,
                 sync_threshold: float = 0.5, use_torch_stream: bool = True, pilots: str = "default",
                 sync_mode: int = SYNC_SCHMIDL_COX, rx_path: int = RX_AUTO, tuning: Optional[dict] = None):
        self.lib = _lib.load()
        if pilots == "stdrng":  # the reference's own tables (restated, unverified) unless explicit tables are given
            sp, st = stdrng_pilots(n_fft)
            preamble = sp if preamble is None else preamble
            training = st if training is None else training
        elif pilots != "default":
            raise OfdmError("pilots must be 'default' or 'stdrng'")
        if not torch.cuda.is_available():
            raise OfdmError("no GPU visible: the OFDM hot path has no CPU fallback")
        self.device = torch.device("cuda", device)
        p = _lib.Params()
        _check(self.lib, self.lib.ofdm_default_params(C.byref(p)), "ofdm_default_params")
        p.n_fft, p.cp_len, p.modulation, p.guard_bands, p.ecc = n_fft, n_fft // 4, modulation, int(guard_bands), ecc
        p.sync_window_reps, p.sync_backoff, p.cfo_mode, p.sync_threshold = (sync_window_reps, sync_backoff, cfo_mode,
                                                                            sync_threshold)
        p.sync_mode = sync_mode
        p.rx_path = rx_path
        self.params = p
        pre = None if preamble is None else np.ascontiguousarray(preamble, dtype=np.complex128)
        trn = None if training is None else np.ascontiguousarray(training, dtype=np.complex128)
        if pre is not None and pre.size != n_fft + n_fft // 4:
            raise OfdmError("preamble must hold n_fft + cp samples")
        if trn is not None and trn.size != n_fft:
            raise OfdmError("training must hold n_fft bins")
        with torch.cuda.device(self.device):
            stream = torch.cuda.current_stream(self.device).cuda_stream if use_torch_stream else 0
        h = C.c_void_p()
        rc = self.lib.ofdm_create(C.byref(p), None if pre is None else pre.ctypes.data,
                                  None if trn is None else trn.ctypes.data, device, C.c_void_p(stream), C.byref(h))
        _check(self.lib, rc, "ofdm_create")
        self.h = h
        self.n_fft, self.cp, self.S = n_fft, n_fft // 4, n_fft + n_fft // 4
        self.modulation, self.guard_bands, self.ecc = modulation, bool(guard_bands), ecc
        self.data_carriers = self.lib.ofdm_data_carriers(h)
        self.bytes_per_symbol = self.lib.ofdm_bytes_per_symbol(h)
        for k, v in {**DEFAULT_TUNING, **(tuning or {})}.items():
            self.set_tuning(k, v)

    def close(self):
        if getattr(self, "h", None):
            self.lib.ofdm_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---------------------------------------------------------------- helpers
    def _ck(self, rc, what):
        _check(self.lib, rc, what, self.h)

    def synchronize(self):
        self._ck(self.lib.ofdm_synchronize(self.h), "ofdm_synchronize")

    def set_tuning(self, key: str, value: int):
        """ofdm_set_tuning: per-context A/B switches and grid shapes (keys in include/ofdm_hip.h)."""
        self._ck(self.lib.ofdm_set_tuning(self.h, key.encode(), int(value)), f"ofdm_set_tuning({key})")

    def get_tuning(self, key: str) -> int:
        v = C.c_int64()
        self._ck(self.lib.ofdm_get_tuning(self.h, key.encode(), C.byref(v)), f"ofdm_get_tuning({key})")
        return int(v.value)

    def last_dispatch(self) -> str:
        """The kernels the last entry point launched, e.g. 'k_rx_prepare+k_rxframe64<finish>' (ofdm_last_dispatch)."""
        buf = C.create_string_buffer(512)
        n = self.lib.ofdm_last_dispatch(self.h, buf, len(buf))
        if n < 0:
            self._ck(n, "ofdm_last_dispatch")
        return buf.value.decode()

    def coded_len(self, payload_bytes: int) -> int:
        return int(self.lib.ofdm_coded_len(self.h, payload_bytes))

    def data_symbols(self, payload_bytes: int) -> int:
        return int(self.lib.ofdm_data_symbols(self.h, payload_bytes))

    def frame_samples(self, payload_bytes: int) -> int:
        return int(self.lib.ofdm_frame_samples(self.h, payload_bytes))

    def to_device(self, a, dtype=None) -> torch.Tensor:
        if isinstance(a, torch.Tensor):
            t = a.to(self.device)
            return t if dtype is None else t.to(dtype)
        arr = np.ascontiguousarray(a)
        if np.iscomplexobj(arr):
            arr = arr.astype(np.complex64)  # sig_to_bytes: f64 -> f32 pairs (src/utils.rs:228-236)
        t = torch.from_numpy(arr).to(self.device)
        return t if dtype is None else t.to(dtype)

    def _cx(self, t: torch.Tensor) -> torch.Tensor:
        if t.dtype != torch.complex64 or not t.is_contiguous() or t.device != self.device:
            raise OfdmError("expected a contiguous complex64 tensor on the context's device")
        return t

    def _u8(self, t: torch.Tensor) -> torch.Tensor:
        if t.dtype != torch.uint8 or not t.is_contiguous() or t.device != self.device:
            raise OfdmError("expected a contiguous uint8 tensor on the context's device")
        return t

    def empty(self, shape, dtype) -> torch.Tensor:
        return torch.empty(shape, dtype=dtype, device=self.device)

    # ---------------------------------------------------------------- stage level
    def fft(self, x: torch.Tensor, inverse: bool = False, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """SignalMut::fft / ifft (src/signals/mod.rs:27-58) over the last dim (== n_fft)."""
        x = self._cx(x)
        assert x.shape[-1] == self.n_fft
        out = torch.empty_like(x) if out is None else self._cx(out)
        self._ck(self.lib.ofdm_fft_batch(self.h, _dev(x), _dev(out), x.numel() // self.n_fft, int(inverse)), "fft")
        return out

    def prefix_block(self, freq: torch.Tensor) -> torch.Tensor:
        """prefix_block::<N, N/4> (src/transmitter.rs:168-181): [..., N] bins -> [..., N+CP] samples."""
        freq = self._cx(freq)
        assert freq.shape[-1] == self.n_fft
        out = self.empty(freq.shape[:-1] + (self.S,), torch.complex64)
        self._ck(self.lib.ofdm_ifft_cp_batch(self.h, _dev(freq), _dev(out), freq.numel() // self.n_fft), "ifft_cp")
        return out

    def tx_symbols(self, data: torch.Tensor, n_sym: Optional[int] = None, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """modulate + encode_block + prefix_block fused (src/transmitter.rs:40-53): a continuous byte stream ->
        [n_sym, n_fft + cp] samples, identical to the three staged calls."""
        data = self._u8(data).reshape(-1)
        nb = data.numel()
        need = (nb + self.bytes_per_symbol - 1) // self.bytes_per_symbol
        n_sym = need if n_sym is None else n_sym
        out = self.empty((n_sym, self.S), torch.complex64) if out is None else self._cx(out)
        assert out.numel() == n_sym * self.S
        self._ck(self.lib.ofdm_tx_symbols_batch(self.h, _dev(data), nb, _dev(out), n_sym), "tx_symbols")
        return out

    def unprefix_block(self, blocks: torch.Tensor) -> torch.Tensor:
        """unprefix_block (src/receiver.rs:99-104): [..., N+CP] samples -> [..., N] bins."""
        blocks = self._cx(blocks)
        assert blocks.shape[-1] == self.S
        out = self.empty(blocks.shape[:-1] + (self.n_fft,), torch.complex64)
        self._ck(self.lib.ofdm_unprefix_batch(self.h, _dev(blocks), _dev(out), blocks.numel() // self.S), "unprefix")
        return out

    def modulate(self, data: torch.Tensor) -> torch.Tensor:
        """modulate (src/transmitter.rs:108-140): uint8[n] -> complex64[ceil(8n/bps)]."""
        data = self._u8(data)
        n = (data.numel() * 8 + self.modulation - 1) // self.modulation
        out = self.empty((n,), torch.complex64)
        self._ck(self.lib.ofdm_qam_map_batch(self.h, _dev(data), data.numel(), _dev(out)), "qam_map")
        return out

    def demodulate(self, sym: torch.Tensor, want_indices: bool = False):
        """demodulate (src/receiver.rs:147-190): complex64[n], n % 8 == 0 -> uint8[n*bps/8]."""
        sym = self._cx(sym)
        n = sym.numel()
        if n % 8 != 0:
            raise OfdmError("demodulate: symbol count must be a multiple of 8 (receiver.rs:153)")
        out = self.empty((n * self.modulation // 8,), torch.uint8)
        idx = self.empty((n,), torch.uint8) if want_indices else None
        self._ck(self.lib.ofdm_qam_demap_batch(self.h, _dev(sym), n, _dev(out), _dev(idx)), "qam_demap")
        return (out, idx) if want_indices else out

    def encode_block(self, data: torch.Tensor) -> torch.Tensor:
        """encode_block (src/transmitter.rs:144-165): [n_sym, data_carriers] -> [n_sym, N]."""
        data = self._cx(data)
        assert data.shape[-1] == self.data_carriers
        out = self.empty(data.shape[:-1] + (self.n_fft,), torch.complex64)
        self._ck(self.lib.ofdm_encode_block_batch(self.h, _dev(data), _dev(out), data.numel() // self.data_carriers),
                 "encode_block")
        return out

    def normalize(self, frames: torch.Tensor) -> torch.Tensor:
        """normalize (src/transmitter.rs:183-194), in place, per row."""
        frames = self._cx(frames)
        f2 = frames.view(-1, frames.shape[-1])
        self._ck(self.lib.ofdm_normalize_batch(self.h, _dev(f2), f2.shape[0], f2.shape[1], f2.shape[1]), "normalize")
        return frames

    def hamming74_encode(self, data: torch.Tensor) -> torch.Tensor:
        data = self._u8(data)
        out = self.empty(((data.numel() + 3) // 4 * 7,), torch.uint8)
        self._ck(self.lib.ofdm_hamming74_encode(self.h, _dev(data), data.numel(), _dev(out)), "hamming74_encode")
        return out

    def hamming74_decode(self, code: torch.Tensor):
        code = self._u8(code)
        out = self.empty((code.numel() // 7 * 4,), torch.uint8)
        fixed = torch.zeros(1, dtype=torch.int32, device=self.device)
        self._ck(self.lib.ofdm_hamming74_decode(self.h, _dev(code), code.numel(), _dev(out), _dev(fixed)),
                 "hamming74_decode")
        return out, fixed

    def sc_correlate(self, frames: torch.Tensor, frame_len: Optional[int] = None, n_lags: int = 0):
        """Schmidl-Cox timing / CFO per row of `frames` [n_frames, stride] -> (d_hat i32, f_delta f64, metric f32)."""
        frames = self._cx(frames)
        f2 = frames.view(-1, frames.shape[-1])
        n, stride = f2.shape
        frame_len = stride if frame_len is None else frame_len
        d = self.empty((n,), torch.int32)
        fd = self.empty((n,), torch.float64)
        m = self.empty((n,), torch.float32)
        self._ck(self.lib.ofdm_sc_correlate_batch(self.h, _dev(f2), n, stride, frame_len, n_lags, _dev(d), _dev(fd),
                                                  _dev(m)), "sc_correlate")
        return d, fd, m

    def xcorr(self, a: torch.Tensor, b: torch.Tensor, want_out: bool = False):
        """SignalRef::xcorr_fft (src/signals/mod.rs:186-217) per row of a [n, len] against b [nb]: (idx_max i32, peak f32
        [, the 2 len - 1 fft_shifted outputs])."""
        a = self._cx(a)
        a2 = a.view(-1, a.shape[-1])
        b = self._cx(b).reshape(-1)
        n, ln = a2.shape
        idx = self.empty((n,), torch.int32)
        pk = self.empty((n,), torch.float32)
        out = self.empty((n, 2 * ln - 1), torch.complex64) if want_out else None
        self._ck(self.lib.ofdm_xcorr_batch(self.h, _dev(a2), n, ln, ln, _dev(b), b.numel(), _dev(idx), _dev(pk), _dev(out),
                                           2 * ln - 1), "xcorr")
        return (idx, pk, out) if want_out else (idx, pk)

    def frequency_correction(self, left_right: torch.Tensor) -> torch.Tensor:
        """frequency_correction (src/receiver.rs:231-240): rows of [left(S) | right(S)] -> f64 |f_delta|."""
        x = self._cx(left_right)
        x2 = x.view(-1, 2 * self.S)
        out = self.empty((x2.shape[0],), torch.float64)
        self._ck(self.lib.ofdm_frequency_correction_batch(self.h, _dev(x2), x2.shape[0], 2 * self.S, self.S, _dev(out)),
                 "frequency_correction")
        return out

    def cfo_rotate(self, frames: torch.Tensor, f_delta: torch.Tensor, first_index: Optional[torch.Tensor] = None):
        """CFO derotation in place (src/receiver.rs:44-50)."""
        frames = self._cx(frames)
        f2 = frames.view(-1, frames.shape[-1])
        assert f_delta.dtype == torch.float64 and f_delta.numel() == f2.shape[0]
        self._ck(self.lib.ofdm_cfo_rotate_batch(self.h, _dev(f2), f2.shape[0], f2.shape[1], f2.shape[1], _dev(f_delta),
                                                _dev(first_index)), "cfo_rotate")
        return frames

    def estimate_channel(self, frames: torch.Tensor, offset: Optional[torch.Tensor] = None,
                         f_delta: Optional[torch.Tensor] = None, frame_len: Optional[int] = None) -> torch.Tensor:
        """estimate_channel (src/receiver.rs:212-229) on the training blocks 5..9 of each frame row."""
        frames = self._cx(frames)
        f2 = frames.view(-1, frames.shape[-1])
        hk = self.empty((f2.shape[0], self.n_fft), torch.complex64)
        self._ck(self.lib.ofdm_estimate_channel_batch(self.h, _dev(f2), f2.shape[0], f2.shape[1],
                                                      f2.shape[1] if frame_len is None else frame_len, _dev(offset),
                                                      _dev(f_delta), _dev(hk)), "estimate_channel")
        return hk

    def rx_demod(self, frames: torch.Tensor, syms_per_frame: int, first_symbol: int = 0,
                 offset: Optional[torch.Tensor] = None, f_delta: Optional[torch.Tensor] = None,
                 hk: Optional[torch.Tensor] = None, frame_len: Optional[int] = None, want_soft: bool = False,
                 out: Optional[torch.Tensor] = None):
        """unprefix_block + equalise + decode_block + demodulate (src/receiver.rs:64-83) per OFDM symbol."""
        frames = self._cx(frames)
        f2 = frames.view(-1, frames.shape[-1])
        n, stride = f2.shape
        nb = syms_per_frame * self.bytes_per_symbol
        out = self.empty((n, nb), torch.uint8) if out is None else self._u8(out)
        soft = self.empty((n, syms_per_frame * self.data_carriers), torch.complex64) if want_soft else None
        hk_stride = 0
        if hk is not None:
            hk = self._cx(hk)
            hk_stride = self.n_fft if hk.dim() == 2 else 0  # [n_frames, N] per frame, [N] shared
            assert hk.shape[-1] == self.n_fft and (hk.dim() == 1 or hk.shape[0] == n)
        self._ck(self.lib.ofdm_rx_demod_batch(self.h, _dev(f2), n, stride, stride if frame_len is None else frame_len,
                                              first_symbol, syms_per_frame, _dev(offset), _dev(f_delta), _dev(hk),
                                              hk_stride, _dev(out), nb, _dev(soft)), "rx_demod")
        return (out, soft) if want_soft else out

    # ---------------------------------------------------------------- pipelines
    def encode_batch(self, payload: torch.Tensor, out: Optional[torch.Tensor] = None,
                     lens: Optional[torch.Tensor] = None) -> torch.Tensor:
        """encode (src/transmitter.rs:11-58) for every row of payload [n_frames, payload_bytes] (uint8).
        lens (int32 [n_frames], optional): true payload length of every row (<= payload_bytes); every frame still has
        the slot size of payload_bytes, its unused tail symbols carry pilots only."""
        payload = self._u8(payload)
        n, nbytes = payload.shape
        frame = self.frame_samples(nbytes)
        out = self.empty((n, frame), torch.complex64) if out is None else self._cx(out)
        if lens is not None:
            lens = lens.to(device=self.device, dtype=torch.int32).contiguous()
            assert lens.numel() == n
        self._ck(self.lib.ofdm_tx_encode_batch(self.h, _dev(payload), n, nbytes, _dev(lens), nbytes, _dev(out), out.shape[-1]),
                 "tx_encode")
        return out

    def decode_batch(self, frames: torch.Tensor, max_symbols: int, n_lags: int = 0, frame_len: Optional[int] = None):
        """decode (src/receiver.rs:9-96) for every row of frames [n_frames, stride]; Schmidl-Cox timing/CFO."""
        frames = self._cx(frames)
        f2 = frames.view(-1, frames.shape[-1])
        n, stride = f2.shape
        ob = max(max_symbols * self.bytes_per_symbol - 16, 4)
        out = self.empty((n, ob), torch.uint8)
        res = {
            "bytes": out, "len": self.empty((n,), torch.int32), "status": self.empty((n,), torch.int32),
            "offset": self.empty((n,), torch.int32), "f_delta": self.empty((n,), torch.float64),
            "metric": self.empty((n,), torch.float32),
        }
        self._ck(self.lib.ofdm_rx_decode_batch(self.h, _dev(f2), n, stride, stride if frame_len is None else frame_len,
                                               n_lags, max_symbols, _dev(out), ob, _dev(res["len"]),
                                               _dev(res["status"]), _dev(res["offset"]), _dev(res["f_delta"]),
                                               _dev(res["metric"])), "rx_decode")
        return res

    def channel_batch(self, tx: torch.Tensor, snr_db: float = 30.0, timing_error: bool = False, seed: int = 1,
                      delay: Optional[torch.Tensor] = None, f_delta: Optional[torch.Tensor] = None,
                      out: Optional[torch.Tensor] = None, span: Optional[int] = None, want_f_delta: bool = False):
        """channel (src/channel.rs:33-74) for every row of tx [n_frames, len]: FIR CHANNEL, optional CFO, pseudo-variance
        noise, seeded (frame f: SplitMix64(seed + f)).  delay (int32) / f_delta (float64) per frame are test-bench
        extensions; span = samples per output row (default len + 63)."""
        tx = self._cx(tx)
        n, ln = tx.shape
        if out is None:
            out = self.empty((n, (ln + 63) if span is None else span), torch.complex64)
        out = self._cx(out)
        assert out.shape[0] == n
        fdo = self.empty((n,), torch.float64) if want_f_delta else None
        if delay is not None:
            delay = delay.to(device=self.device, dtype=torch.int32).contiguous()
        if f_delta is not None:
            f_delta = f_delta.to(device=self.device, dtype=torch.float64).contiguous()
        self._ck(self.lib.ofdm_channel_batch(self.h, _dev(tx), n, ln, ln, float(snr_db), int(timing_error), int(seed) & (2 ** 64 - 1),
                                             _dev(delay), _dev(f_delta), _dev(out), out.shape[-1], out.shape[-1], _dev(fdo)),
                 "channel")
        return (out, fdo) if want_f_delta else out

    # ---------------------------------------------------------------- one long capture (examples/jetson_rx.rs:15-17,84-86)
    def sc_correlate_long(self, capture: torch.Tensor, lag_lo: int = 0, lag_hi: int = 0, slice_lags: int = 0):
        """ofdm_sc_correlate_long: the Schmidl-Cox detection of ONE capture whose first crossing lies in [lag_lo, lag_hi)
        (lag_hi = 0: up to the last lag), searched as a batch of overlapping slices -> (d_hat or -1, f_delta, metric)."""
        x = self._cx(capture).reshape(-1)
        d, fd, m = C.c_int64(), C.c_double(), C.c_float()
        self._ck(self.lib.ofdm_sc_correlate_long(self.h, _dev(x), x.numel(), lag_lo, lag_hi, slice_lags, C.byref(d), C.byref(fd),
                                                 C.byref(m)), "sc_correlate_long")
        return int(d.value), float(fd.value), float(m.value)

    def decode_long(self, capture: torch.Tensor, max_symbols: int, lag_lo: int = 0, lag_hi: int = 0, d_hat_known: int = -1):
        """ofdm_rx_decode_long: decode (src/receiver.rs:9-96) of ONE long capture on the device -> dict(bytes, len, status, offset,
        f_delta, metric), the result of decode_batch on the whole capture as a single frame."""
        x = self._cx(capture).reshape(-1)
        ob = max(max_symbols * self.bytes_per_symbol - 16, 4)
        out = self.empty((ob,), torch.uint8)
        ln, st, off, fd, m = C.c_int32(), C.c_int32(), C.c_int64(), C.c_double(), C.c_float()
        self._ck(self.lib.ofdm_rx_decode_long(self.h, _dev(x), x.numel(), lag_lo, lag_hi, d_hat_known, max_symbols, _dev(out), ob,
                                              C.byref(ln), C.byref(st), C.byref(off), C.byref(fd), C.byref(m)), "rx_decode_long")
        return {"bytes": out, "len": int(ln.value), "status": int(st.value), "offset": int(off.value), "f_delta": float(fd.value),
                "metric": float(m.value)}

    def decode_long_host(self, capture: np.ndarray, max_symbols: int):
        """ofdm_rx_decode_long_host: the same from a host array of complex64 (pinned_empty() memory is DMA-ed in place)."""
        x = np.ascontiguousarray(capture, dtype=np.complex64).reshape(-1)
        ob = max(max_symbols * self.bytes_per_symbol - 16, 4)
        out = np.zeros(ob, np.uint8)
        ln, st, off, fd, m = C.c_int32(), C.c_int32(), C.c_int64(), C.c_double(), C.c_float()
        self._ck(self.lib.ofdm_rx_decode_long_host(self.h, _host(x), x.size, max_symbols, _host(out), ob, C.byref(ln), C.byref(st),
                                                   C.byref(off), C.byref(fd), C.byref(m)), "rx_decode_long_host")
        return {"bytes": out, "len": int(ln.value), "status": int(st.value), "offset": int(off.value), "f_delta": float(fd.value),
                "metric": float(m.value)}

    # ---------------------------------------------------------------- host buffers (H2D / kernels / D2H pipelined in the library)
    def decode_host(self, frames: np.ndarray, max_symbols: int, n_lags: int = 0, frame_len: Optional[int] = None,
                    chunk_frames: int = 0, out: Optional[dict] = None):
        """ofdm_rx_decode_host: decode_batch for a host array [n_frames, stride] of complex64 -> dict of host arrays."""
        x = frames if frames.dtype == np.complex64 and frames.flags.c_contiguous else np.ascontiguousarray(frames, dtype=np.complex64)
        n, stride = x.shape
        ob = max(max_symbols * self.bytes_per_symbol - 16, 4)
        res = out or {"bytes": np.zeros((n, ob), np.uint8), "len": np.zeros(n, np.int32), "status": np.zeros(n, np.int32),
                      "offset": np.zeros(n, np.int32), "f_delta": np.zeros(n, np.float64), "metric": np.zeros(n, np.float32)}
        self._ck(self.lib.ofdm_rx_decode_host(self.h, _host(x), n, stride, stride if frame_len is None else frame_len, n_lags,
                                              max_symbols, _host(res["bytes"]), res["bytes"].shape[1], _host(res["len"]),
                                              _host(res["status"]), _host(res["offset"]), _host(res["f_delta"]), _host(res["metric"]),
                                              chunk_frames), "rx_decode_host")
        return res

    def demod_host(self, frames: np.ndarray, syms_per_frame: int, first_symbol: int = 0, chunk_frames: int = 0,
                   out: Optional[np.ndarray] = None) -> np.ndarray:
        """ofdm_rx_demod_host: rx_demod of regular streams for a host array [n_frames, stride] -> uint8 [n_frames, bytes]."""
        x = frames if frames.dtype == np.complex64 and frames.flags.c_contiguous else np.ascontiguousarray(frames, dtype=np.complex64)
        n, stride = x.shape
        nb = syms_per_frame * self.bytes_per_symbol
        out = np.zeros((n, nb), np.uint8) if out is None else out
        self._ck(self.lib.ofdm_rx_demod_host(self.h, _host(x), n, stride, stride, first_symbol, syms_per_frame, _host(out),
                                             out.shape[1], chunk_frames), "rx_demod_host")
        return out

    def encode_host(self, payload: np.ndarray, lens: Optional[np.ndarray] = None, chunk_frames: int = 0,
                    out: Optional[np.ndarray] = None) -> np.ndarray:
        """ofdm_tx_encode_host: encode_batch for a host array [n_frames, payload_bytes] of uint8 -> complex64 [n_frames, frame]."""
        pay = np.ascontiguousarray(payload, dtype=np.uint8)
        n, nbytes = pay.shape
        frame = self.frame_samples(nbytes)
        out = np.zeros((n, frame), np.complex64) if out is None else out
        if lens is not None:
            lens = np.ascontiguousarray(lens, dtype=np.int32)
        self._ck(self.lib.ofdm_tx_encode_host(self.h, _host(pay), n, nbytes, _host(lens), nbytes, _host(out), out.shape[1],
                                              chunk_frames), "tx_encode_host")
        return out

    def use_own_stream(self):
        """ofdm_use_own_stream: a non-blocking stream owned by this context (several contexts side by side on one device)."""
        self._ck(self.lib.ofdm_use_own_stream(self.h), "ofdm_use_own_stream")

    def hbm_read_probe(self, samples: torch.Tensor, pattern: int = 0):
        """Measurement helper: read-only pass over a buffer of 80-sample symbols in the demod kernel's access pattern (0),
        over whole symbols (1) or with unit-stride 16-byte loads (2)."""
        samples = self._cx(samples)
        self._ck(self.lib.ofdm_hbm_read_probe(self.h, _dev(samples), samples.numel() // 80, pattern), "hbm_read_probe")

    # event timing on the context's stream (bench.py)
    def timer_start(self):
        self._ck(self.lib.ofdm_timer_start(self.h), "timer_start")

    def timer_stop_ms(self) -> float:
        ms = C.c_float()
        self._ck(self.lib.ofdm_timer_stop_ms(self.h, C.byref(ms)), "timer_stop")
        return float(ms.value)


# -------------------------------------------------------------------- reference-shaped free functions
_CTX_CACHE = {}


def _ctx(n_fft, modulation, guard_bands, ecc=ECC_NONE, _replica: int = 0, **kw) -> Context:
    """One cached context per parameter set (and replica index): the free functions never pay ofdm_create -- table uploads,
    workspace growth -- more than once."""
    key = (n_fft, modulation, bool(guard_bands), ecc, _replica, tuple(sorted(kw.items())))
    if key not in _CTX_CACHE:
        _CTX_CACHE[key] = Context(n_fft=n_fft, modulation=modulation, guard_bands=guard_bands, ecc=ecc, **kw)
    return _CTX_CACHE[key]


def encode(data: bytes, guard_bands: Optional[bool] = None, modulation: Optional[int] = None, n_fft: int = 64,
           ecc: int = ECC_NONE, **kw) -> np.ndarray:
    """`ofdm::encode!(data, guard_bands, modulation)` (src/transmitter.rs:10-58): bytes -> Vec<Complex64>.
    Defaults as the reference: guard_bands=false, modulation=Bpsk."""
    ctx = _ctx(n_fft, BPSK if modulation is None else modulation, bool(guard_bands), ecc, **kw)
    pay = torch.frombuffer(bytearray(data) if len(data) else bytearray(1), dtype=torch.uint8)[: len(data)]
    pay = pay.reshape(1, len(data)).to(ctx.device)
    frames = ctx.encode_batch(pay)
    ctx.synchronize()
    return frames[0].cpu().numpy().astype(np.complex128)


def channel(transmission, snr: Optional[float] = None, timing_error: Optional[bool] = None, seed: int = 1) -> np.ndarray:
    """`ofdm::channel!(transmission, snr, timing_error)` (src/channel.rs:32-74) on the GPU, seeded (the reference's
    thread_rng is not reproducible).  Defaults as the reference: snr 30 dB, timing_error false."""
    ctx = _ctx(64, BPSK, False)
    x = ctx.to_device(np.asarray(transmission)).reshape(1, -1)
    y = ctx.channel_batch(x, 30.0 if snr is None else snr, bool(timing_error), seed)
    ctx.synchronize()
    return y[0].cpu().numpy().astype(np.complex128)


def decode(samples, guard_bands: Optional[bool] = None, modulation: Optional[int] = None, n_fft: int = 64,
           ecc: int = ECC_NONE, **sync) -> bytes:
    """`ofdm::decode!(samples, guard_bands, modulation)` (src/receiver.rs:8-96): Vec<Complex64> -> Result<Vec<u8>>.
    Raises DecodeError("Input not long enough, bailing early") where the reference returns Err."""
    ctx = _ctx(n_fft, BPSK if modulation is None else modulation, bool(guard_bands), ecc, **sync)
    x = ctx.to_device(np.asarray(samples)).reshape(1, -1)
    max_symbols = max((x.shape[1] + ctx.S - 1) // ctx.S - 10, 1)
    res = ctx.decode_batch(x, max_symbols=max_symbols)
    ctx.synchronize()
    status = int(res["status"][0])
    if status == FRAME_SHORT:
        raise DecodeError("Input not long enough, bailing early")
    if status != FRAME_OK:
        raise DecodeError({FRAME_NOSYNC: "no preamble found", FRAME_HEADER: "no length header decoded",
                           FRAME_BADTIMING: "timing offset outside the capture (the reference panics in split_off)"}.get(status, "decode failed"))
    n = int(res["len"][0])
    return bytes(res["bytes"][0, :n].cpu().numpy())


def decode_long(samples, guard_bands: Optional[bool] = None, modulation: Optional[int] = None, n_fft: int = 64,
                ecc: int = ECC_NONE, world: int = 1, max_symbols: Optional[int] = None, device: int = 0, **sync):
    """`ofdm::decode!` of ONE long capture (the 2 M-sample buffers of examples/jetson_rx.rs:15-17,48-49,84-86) on the HIP path.
    world > 1 rehearses the multi-GPU halo split on one device: `world` contexts, context r searching the lags
    dist.lag_ranges(...)[r] of the shared capture (ofdm_sc_correlate_long), the lowest range with a detection wins
    (dist.merge_first_detection), and that context runs the receive chain (ofdm_rx_decode_long with the merged detection).
    Returns dict(bytes, len, status, offset, f_delta, metric); raises DecodeError where the reference returns Err."""
    from . import dist

    mod = BPSK if modulation is None else modulation
    ctxs = [_ctx(n_fft, mod, bool(guard_bands), ecc, device=device, _replica=r, **sync) for r in range(max(world, 1))]
    c0 = ctxs[0]
    x = samples if isinstance(samples, torch.Tensor) else c0.to_device(np.asarray(samples))
    x = x.reshape(-1)
    if max_symbols is None:
        max_symbols = max((x.numel() + c0.S - 1) // c0.S - 10, 1)
    if world <= 1 or c0.params.sync_mode != SYNC_SCHMIDL_COX:
        res = c0.decode_long(x, max_symbols)
    else:
        L, W = c0.S, c0.params.sync_window_reps * c0.S
        dets = []
        for (lo, hi), c in zip(dist.lag_ranges(x.numel(), world, L, W), ctxs):
            d, fd, m = c.sc_correlate_long(x, lo, hi) if hi > lo else (-1, 0.0, 0.0)
            dets.append((0, hi, d, fd, m))  # d is already a lag of the whole capture
        d, fd, m = dist.merge_first_detection(dets)
        winner = next((c for c, det in zip(ctxs, dets) if det[2] >= 0), c0)
        res = winner.decode_long(x, max_symbols, d_hat_known=d) if d >= 0 else {
            "bytes": c0.empty((4,), torch.uint8), "len": 0, "status": FRAME_NOSYNC, "offset": 0, "f_delta": 0.0, "metric": 0.0}
    if res["status"] == FRAME_SHORT:
        raise DecodeError("Input not long enough, bailing early")
    return res
