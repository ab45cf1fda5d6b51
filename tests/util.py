"""Shared helpers for the parity tests: seeded synthetic inputs and comparison rules.

Inputs handed to the GPU are fc32 (the wire format, src/utils.rs:228-254); the oracle is fed the SAME
f32-rounded samples widened to f64 (bytes_to_sig), so both sides see identical data.
"""
import numpy as np

TAPS = None


def fc32(x):
    """Round complex128 to the fc32 wire format and widen back (what both GPU and oracle consume)."""
    return np.asarray(x).astype(np.complex64)


def wide(x32):
    return np.asarray(x32).astype(np.complex128)


def rel_err(a, b):
    """norm-relative error (per-bin relative error is unbounded at null carriers)."""
    a = np.asarray(a).astype(np.complex128).ravel()
    b = np.asarray(b).astype(np.complex128).ravel()
    n = np.linalg.norm(b)
    return float(np.linalg.norm(a - b) / (n if n > 0 else 1.0))


def awgn(rng, shape, sigma):
    return sigma * (rng.standard_normal(shape) + 1j * rng.standard_normal(shape))


def make_symbols(orc, rng, n_sym, n_fft, guard, mod, snr_db=30.0):
    """Config-2 style input: n_sym OFDM symbols (CP + N) carrying random points, AWGN, H == 1.
    Returns (samples complex64 [n_sym*S], payload bytes)."""
    nd = orc.data_carriers(n_fft, guard)
    nbytes = n_sym * nd * mod // 8
    data = bytes(rng.integers(0, 256, nbytes, dtype=np.uint8))
    pts = orc.modulate(data, mod)
    out = []
    for s in range(n_sym):
        blk, used = orc.encode_block(pts[s * nd:(s + 1) * nd], n_fft, guard)
        out.append(orc.prefix_block(blk))
    x = np.concatenate(out)
    p = np.mean(np.abs(x) ** 2)
    x = x + awgn(rng, x.shape, np.sqrt(p / 10 ** (snr_db / 10) / 2))
    return fc32(x), data


def make_symbols_np(orc, rng, n_sym, n_fft, guard, mod, snr_db=30.0):
    """make_symbols for large batches: the same construction (modulate -> encode_block -> prefix_block -> AWGN) with the
    per-symbol oracle calls replaced by numpy array operations (the carrier map comes from the oracle's carrier_class,
    the points from the oracle's modulate).  Only an INPUT generator: expected outputs always come from the oracle."""
    nd = orc.data_carriers(n_fft, guard)
    nbytes = n_sym * nd * mod // 8
    data = bytes(rng.integers(0, 256, nbytes, dtype=np.uint8))
    pts = np.asarray(orc.modulate(data, mod)).reshape(n_sym, nd)
    cls = np.array([orc.carrier_class(i, n_fft, guard) for i in range(n_fft)])
    bins = np.zeros((n_sym, n_fft), np.complex128)
    bins[:, cls == 0] = pts
    bins[:, cls == 2] = 1.0
    t = np.fft.ifft(bins, axis=1)
    cp = n_fft // 4
    x = np.concatenate([t[:, n_fft - cp:], t], axis=1).reshape(-1)
    p = np.mean(np.abs(x) ** 2)
    x = x + awgn(rng, x.shape, np.sqrt(p / 10 ** (snr_db / 10) / 2))
    return fc32(x), data


def through_channel(orc, rng, tx, span, delay, f_delta, snr_db=30.0, taps=True, data_start=None):
    """Config-3 style capture: `delay` leading zeros, FIR CHANNEL (src/channel.rs:26-31), CFO
    exp(+j f (i+1)) (channel.rs:58-62), AWGN, cut/padded to `span` samples.
    The SNR is set against the power of the DATA symbols (tx[data_start:]): the lock / preamble blocks are
    time-domain constants while IFFT output shrinks as 1/N, so for large N the header would dominate."""
    y = np.convolve(tx, orc.channel_taps())[: tx.size + 24] if taps else tx.copy()
    buf = np.zeros(span, np.complex128)
    n = min(span - delay, y.size)
    buf[delay:delay + n] = y[:n]
    buf *= np.exp(1j * f_delta * np.arange(1, span + 1))
    p = np.mean(np.abs(y[data_start or 0:]) ** 2)
    if snr_db is not None:
        buf += awgn(rng, span, np.sqrt(p / 10 ** (snr_db / 10) / 2))
    return fc32(buf)


def decision_margin(soft, mod):
    """Distance of every soft point to its nearest hard-decision boundary, per axis (in the same units as the
    points).  Used to excuse a differing decision ONLY where the oracle's own value sits within the
    floating-point tolerance of a boundary."""
    soft = np.asarray(soft)
    if mod == 1:
        return np.abs(soft.real)
    m = mod // 2
    M = 1 << m
    def axis(x):
        u = x * (M - 1) / 2.0          # boundaries at integers (inner ones only matter up to the clamp)
        d = np.abs(u - np.round(u))
        outer = np.abs(u) > (M / 2 - 0.5)  # beyond the outermost boundary: clamped, no boundary nearby
        return np.where(outer, np.inf, d) * 2.0 / (M - 1)
    return np.minimum(axis(soft.real), axis(soft.imag))


def assert_bytes_match(got: bytes, want: bytes, soft, mod, nd_per_byte_group=None, tol=1e-5, what=""):
    """Bit-exact byte equality, excusing only decisions whose oracle soft value lies within `tol` (relative to
    the constellation scale 1.0) of a boundary.  Returns the number of excused points (normally 0)."""
    assert len(got) == len(want), f"{what}: length {len(got)} != {len(want)}"
    if got == want:
        return 0
    g = np.unpackbits(np.frombuffer(got, np.uint8), bitorder="little")
    w = np.unpackbits(np.frombuffer(want, np.uint8), bitorder="little")
    bad_pts = np.unique(np.nonzero(g != w)[0] // mod)
    margin = decision_margin(np.asarray(soft)[bad_pts], mod)
    assert np.all(margin < tol), f"{what}: {np.sum(margin >= tol)} decisions differ away from any boundary"
    return int(bad_pts.size)
