"""GPU parity of the host-buffer and long-capture entry points (include/ofdm_hip.h: "host buffers", "one long capture").

The reference's decode owns ONE host Vec of samples and looks for the packet anywhere in it (src/receiver.rs:9-25); its receiver
example hands it 2 000 000-sample buffers (examples/jetson_rx.rs:15-17,48-49,84-86).  Checked here:
  * ofdm_sc_correlate_long / ofdm_rx_decode_long[_host] against the oracle's search / decode of the WHOLE capture (threshold-then-peak
    must give the same lag whether the capture is searched as one frame, as a batch of overlapping slices, or split over `world`
    contexts), with the packet early, straddling a cut, late, and absent;
  * ofdm_rx_decode_host / ofdm_rx_demod_host / ofdm_tx_encode_host against the device-buffer entry points they pipeline: byte for
    byte, for pageable and pinned memory, with chunk sizes that wrap the three slots and leave a ragged last chunk.
"""
import numpy as np
import pytest

from util import rel_err, through_channel, wide

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def api(ofdm):
    import torch

    assert torch.cuda.is_available(), "GPU tests need a GPU"
    from ofdm_amd import api as _api

    return _api


def _capture(orc, rng, n, starts, n_fft=64, mod=6, guard=True, nbytes=560, snr_db=30.0, fd=None):
    """One capture of n samples with a frame at each of `starts` (FIR channel, CFO, noise everywhere)."""
    S = n_fft + n_fft // 4
    cap = np.zeros(n, np.complex64)
    pays = []
    sigma = None
    for st in starts:
        pay = bytes(rng.integers(0, 256, nbytes, dtype=np.uint8))
        tx = orc.encode(pay, guard, mod, n_fft)
        f = (rng.random() * 1.8 - 0.9) * np.pi / S if fd is None else fd
        seg_len = tx.size + 64
        seg = through_channel(orc, rng, tx, seg_len, 0, f, snr_db=None, data_start=10 * S)
        # the channel's CFO phase counts from the segment's first sample: any start is a constant phase, which the receiver removes
        cap[st:st + seg_len] += seg[: max(0, min(seg_len, n - st))]
        if sigma is None:
            p = float(np.mean(np.abs(tx[10 * S:]) ** 2))
            sigma = np.sqrt(p / 10 ** (snr_db / 10) / 2)
        pays.append(pay)
    if sigma is None:
        sigma = 0.01
    cap += (sigma * (rng.standard_normal(n) + 1j * rng.standard_normal(n))).astype(np.complex64)
    return cap, pays


def _same_decode(got, want, what):
    assert got["status"] == want["status"], (what, got["status"], want["status"])
    if want["status"] != 0:
        assert got["len"] == 0, what
        return
    assert got["offset"] == want["offset"], (what, got["offset"], want["offset"])
    assert abs(got["f_delta"] - want["f_delta"]) <= 1e-9, what
    gb = got["bytes"][: got["len"]]
    gb = bytes(gb.cpu().numpy()) if hasattr(gb, "cpu") else bytes(gb)
    assert gb == want["bytes"], what


def test_decode_long_two_million_samples(api, orc):
    """the reference's operating point -- ONE decode! per 2 000 000-sample buffer -- on the HIP path, against
    orc.decode_sc on the whole capture: frame early, straddling a cut of the 2 / 4 / 7-way lag split, late, absent; world = 1
    (slices only) and world = 2, 4, 7 contexts on the one device (ofdm_sc_correlate_long per lag range, merge, ofdm_rx_decode_long)."""
    from ofdm_amd import dist

    rng = np.random.default_rng(404)
    n, L, W = 2_000_000, 80, 240
    D = 16
    cuts = {w: dist.lag_ranges(n, w, L, W) for w in (2, 4, 7)}
    cases = [("early", [4000]), ("straddles the 2-way cut", [cuts[2][0][1] - 150]), ("straddles a 7-way cut", [cuts[7][2][1] - 95]),
             ("crossing in the overrun of a 4-way range", [cuts[4][0][1] - 60]), ("two frames", [cuts[4][2][1] + 20_000, 300_000]),
             ("late", [n - 2080 - 64 - 10]), ("cut off by the end of the capture: clipped peak window, too short to decode", [n - 500]),
             ("absent", [])]
    ctx = api.Context(n_fft=64, modulation=api.QAM64, guard_bands=True)
    for what, starts in cases:
        cap, pays = _capture(orc, rng, n, starts)
        want = orc.decode_sc(wide(cap), True, orc.QAM64, 64, max_symbols=D)
        wd, _, wm, wfd = orc.sc_sync(wide(cap), L, 3, 0, 0.5)
        x = ctx.to_device(cap)
        # the search alone, as slices of one context
        d, fd, m = ctx.sc_correlate_long(x)
        assert d == wd, (what, d, wd)
        if wd >= 0:
            assert abs(fd - wfd) <= 1e-9 and abs(m - wm) <= 1e-6, what
            assert "k_sc80" in ctx.last_dispatch(), ctx.last_dispatch()     # the one-tile detector serves the slices (no fallback)
        for world in (1, 2, 4, 7):
            try:
                got = api.decode_long(x, True, api.QAM64, world=world, max_symbols=D)
            except api.DecodeError as e:      # the reference's Err: "Input not long enough, bailing early" (src/receiver.rs:27-29)
                assert "not long enough" in str(e)
                got = {"status": api.FRAME_SHORT, "len": 0}
            _same_decode(got, want, (what, world))
        # host entry point: pageable and pinned memory
        got = ctx.decode_long_host(cap, D)
        _same_decode(got, want, (what, "host pageable"))
    pin = api.pinned_empty((n,), np.complex64)
    pin[:] = cap
    assert ctx.lib.ofdm_host_is_pinned(pin.ctypes.data, pin.nbytes) == 1 and ctx.lib.ofdm_host_is_pinned(cap.ctypes.data, cap.nbytes) == 0
    _same_decode(ctx.decode_long_host(pin, D), want, "host pinned")


def test_decode_long_from_a_lag_in_front_of_the_second_packet(api, orc):
    """ofdm_rx_decode_long(lag_lo = end of packet 1) on a two-packet capture.  The second frame's trimmed start and the
    sub-capture the chain used to search both lie in front of lag_lo, where packet 1's tail still crosses the threshold: the answer must
    be the one of decode on the capture from lag_lo on (re-based), not packet 1's tail.  lag_lo swept over the tail of packet 1, odd
    and even, and over a point inside packet 2's plateau rise."""
    rng = np.random.default_rng(417)
    n, D = 40_000, 16
    st1, st2 = 3000, 3000 + 2080 + 64 + 170           # packet 2 starts 170 samples after packet 1's channel tail
    cap, pays = _capture(orc, rng, n, [st1, st2])
    ctx = api.Context(n_fft=64, modulation=api.QAM64, guard_bands=True)
    x = ctx.to_device(cap)
    whole = orc.decode_sc(wide(cap), True, orc.QAM64, 64, max_symbols=D)
    assert whole["status"] == 0 and bytes(whole["bytes"]) == pays[0]
    d1 = orc.sc_sync(wide(cap), 80, 3, 0, 0.5)[0]
    seen_second = 0
    for lag_lo in (d1 + 1, d1 + 200, st1 + 2080 - 150, st1 + 2080 - 41, st1 + 2080 + 30, st2 - 75, st2 + 21):
        tail = wide(cap[lag_lo:])
        wd, _, wm, wfd = orc.sc_sync(tail, 80, 3, 0, 0.5)
        assert wd >= 0
        d, fd, m = ctx.sc_correlate_long(x, lag_lo=lag_lo)
        assert d == lag_lo + wd and abs(fd - wfd) <= 1e-9, (lag_lo, d, lag_lo + wd)
        got = ctx.decode_long(x, D, lag_lo=lag_lo)
        # the oracle on the capture from lag_lo on decides WHICH crossing and its CFO; the frame it belongs to starts L + backoff samples
        # in front of the peak -- possibly in front of lag_lo --, so the receive chain is the oracle's decode_given on the whole capture
        off = max(lag_lo + wd - 80 - 4, 0)
        want = orc.decode_given(wide(cap), off, wfd, True, orc.QAM64, 64, max_symbols=D)
        assert got["status"] == want["status"], (lag_lo, got["status"], want["status"])
        if want["status"] == 0:
            assert got["offset"] == off and abs(got["f_delta"] - wfd) <= 1e-9, (lag_lo, got["offset"], off)
            assert bytes(got["bytes"][: got["len"]].cpu().numpy()) == want["bytes"], lag_lo
            seen_second += sum(a != b for a, b in zip(want["bytes"], pays[1])) <= 4   # packet 2 (64-QAM at 30 dB: a byte or two may be wrong)
    assert seen_second >= 3


@pytest.mark.parametrize("n_fft,reps", [(64, 3), (64, 1), (256, 3), (1024, 2), (4096, 3)])
def test_sc_correlate_long_slices_and_lag_ranges(api, orc, n_fft, reps):
    """The slice batch against one search of the whole capture for every period family (L = 80: k_sc_cf tiles; L >= 160:
    k_sc_stream), with slices small enough that the crossing is swept across slice cuts, the tail frame and the capture's end;
    explicit lag ranges must return the detection of THAT range (what a rank of the multi-GPU split asks for)."""
    rng = np.random.default_rng(n_fft + reps)
    S = n_fft + n_fft // 4
    L, W = S, reps * S
    ctx = api.Context(n_fft=n_fft, modulation=api.QAM16, guard_bands=True, sync_window_reps=reps)
    nbytes = 120 * (n_fft // 64)
    flen = ctx.frame_samples(nbytes)
    n = flen + 14 * (2 * W + L) + 1000
    own = 2 * W + L + 2 * (S // 8)                     # small slices: ~14 of them, halo = half of every frame read
    for k in range(9):
        st = int(rng.integers(0, n - flen + flen // 3))      # the last third of the draws cuts the frame off at the capture's end
        if k == 7:
            st = n - flen - 64                               # peak window clipped by the last lag
        starts = [] if k == 8 else [st]
        cap, _ = _capture(orc, rng, n, starts, n_fft=n_fft, mod=4, nbytes=nbytes, snr_db=28.0)
        wd, _, wm, wfd = orc.sc_sync(wide(cap), L, reps, 0, 0.5)
        x = ctx.to_device(cap)
        for sl in (0, own, own + 2 * S + 6):
            d, fd, m = ctx.sc_correlate_long(x, slice_lags=sl)
            assert d == wd, (n_fft, reps, k, sl, d, wd)
            if wd >= 0:
                assert abs(fd - wfd) <= 1e-9 and abs(m - wm) <= 1e-6
        # lag ranges: [0, c) and [c, end) around the crossing
        if wd >= 0:
            for c in (max(wd - W, 2) & ~1, (wd + 40) & ~1):
                lo_part = ctx.sc_correlate_long(x, 0, c, slice_lags=own)
                hi_part = ctx.sc_correlate_long(x, c, 0, slice_lags=own)
                first = lo_part if lo_part[0] >= 0 else hi_part
                assert first[0] == wd, (n_fft, reps, k, c, lo_part, hi_part, wd)
    # a capture without a single valid lag
    short = ctx.to_device(np.zeros(W + L - 2, np.complex64))
    assert ctx.sc_correlate_long(short)[0] == -1


def test_decode_long_other_lengths_and_reference_sync(api, orc):
    """ofdm_rx_decode_long for N = 1024 (k_sc_stream slices, k_rxframe1024) and with the reference's own detector
    (OFDM_SYNC_REFERENCE: xcorr_fft over the whole capture, src/receiver.rs:20-25) == decode_batch of the capture as one frame."""
    rng = np.random.default_rng(91)
    ctx = api.Context(n_fft=1024, modulation=api.QAM64, guard_bands=True, ecc=api.ECC_HAMMING74)
    nbytes = 1304
    flen = ctx.frame_samples(nbytes)
    D = ctx.data_symbols(nbytes)
    n = 300_000
    for st in (1000, 150_001, n - flen - 40):
        pay = bytes(rng.integers(0, 256, nbytes, dtype=np.uint8))
        tx = orc.encode(orc.hamming74_encode(pay), True, orc.QAM64, 1024)
        cap = np.zeros(n, np.complex64)
        seg = through_channel(orc, rng, tx, tx.size + 64, 0, 0.0007, snr_db=None, data_start=12800)
        cap[st:st + seg.size] = seg[: n - st]
        cap += (2e-4 * (rng.standard_normal(n) + 1j * rng.standard_normal(n))).astype(np.complex64)
        x = ctx.to_device(cap)
        got = ctx.decode_long(x, D)
        one = ctx.decode_batch(x.reshape(1, -1), max_symbols=D)
        ctx.synchronize()
        assert got["status"] == int(one["status"][0]) == 0 and got["offset"] == int(one["offset"][0])
        assert got["len"] == int(one["len"][0]) and bool((got["bytes"][: got["len"]] == one["bytes"][0, : got["len"]]).all())
        assert abs(got["f_delta"] - float(one["f_delta"][0])) <= 1e-12
        w = orc.decode_sc(wide(cap), True, orc.QAM64, 1024, max_symbols=D)
        assert got["offset"] == w["offset"] and bytes(got["bytes"][: got["len"]].cpu().numpy()) == orc.hamming74_decode(w["bytes"])[0]
    ref = api.Context(n_fft=64, modulation=api.QPSK, sync_mode=api.SYNC_REFERENCE)
    pay = bytes(rng.integers(0, 256, 400, dtype=np.uint8))
    cap, _ = _capture(orc, rng, 50_000, [31_007], mod=2, guard=False, nbytes=400, fd=0.004)
    x = ref.to_device(cap)
    got = ref.decode_long(x, 30)
    one = ref.decode_batch(x.reshape(1, -1), max_symbols=30)
    ref.synchronize()
    assert got["status"] == int(one["status"][0]) and got["offset"] == int(one["offset"][0]) and got["len"] == int(one["len"][0])
    assert bool((got["bytes"][: got["len"]] == one["bytes"][0, : got["len"]]).all())
    with pytest.raises(api.OfdmError):  # a partial lag range has no meaning for an argmax over the whole capture
        ref.decode_long(x, 30, lag_lo=100)


def test_decode_long_with_the_reference_symbol_count(api, orc):
    """The reference demodulates EVERY chunk from the trimmed start to the end of the capture and lets the length header truncate
    (src/receiver.rs:54-95); max_symbols = None asks for exactly that ((n + S - 1) / S - 10 symbols: 1 240 here).  Device and host
    entry points against the oracle with no symbol limit, and the C-level free function `api.decode` on the same capture."""
    rng = np.random.default_rng(808)
    n = 100_000
    for mod, guard, nbytes, st in ((api.QAM64, True, 560, 30_011), (api.QPSK, False, 400, 77_000)):
        cap, pays = _capture(orc, rng, n, [st], mod=mod, guard=guard, nbytes=nbytes)
        want = orc.decode_sc(wide(cap), guard, mod, 64)           # max_symbols = 0: every symbol the capture holds
        assert want["status"] == 0 and len(want["bytes"]) == nbytes
        got = api.decode_long(cap, guard, mod)
        _same_decode(got, want, ("device", mod))
        ctx = api.Context(n_fft=64, modulation=mod, guard_bands=guard)
        _same_decode(ctx.decode_long_host(cap, (n + 79) // 80 - 10), want, ("host", mod))
        assert api.decode(cap, guard, mod) == want["bytes"]


@pytest.mark.parametrize("pinned", [False, True])
def test_host_pipelines_equal_the_device_entry_points(api, orc, pinned):
    """ofdm_rx_decode_host / ofdm_rx_demod_host / ofdm_tx_encode_host == the device-buffer calls they wrap, byte for byte:
    chunk sizes that wrap the three slots several times, a ragged last chunk, one chunk, and the library's own choice."""
    import torch

    rng = np.random.default_rng(55 + pinned)
    ctx = api.Context(n_fft=64, modulation=api.QAM64, guard_bands=True)
    nfr, nbytes, span = 53, 560, 2176
    alloc = (lambda shape, dt: api.pinned_empty(shape, dt)) if pinned else (lambda shape, dt: np.zeros(shape, dt))
    # --- encode: ragged lengths
    pay = alloc((nfr, nbytes), np.uint8)
    pay[:] = rng.integers(0, 256, (nfr, nbytes), dtype=np.uint8)
    lens = rng.integers(0, nbytes + 1, nfr).astype(np.int32)
    lens[:3] = (0, nbytes, 1)
    want_tx = ctx.encode_batch(torch.from_numpy(np.ascontiguousarray(pay)).to(ctx.device), lens=torch.from_numpy(lens)).cpu().numpy()
    for chunk in (5, 16, nfr, 0):
        out = alloc(want_tx.shape, np.complex64)
        got = ctx.encode_host(pay, lens=lens, chunk_frames=chunk, out=out)
        assert np.array_equal(got.view(np.uint32), want_tx.view(np.uint32)), ("encode", chunk)
    over = lens.copy(); over[1] = nbytes + 100; over[4] = 2 ** 30            # lengths above payload_bytes are clamped, on both sides alike
    want_over = ctx.encode_batch(torch.from_numpy(np.ascontiguousarray(pay)).to(ctx.device), lens=torch.from_numpy(over)).cpu().numpy()
    clamped = over.copy(); clamped[[1, 4]] = nbytes
    want_clamped = ctx.encode_batch(torch.from_numpy(np.ascontiguousarray(pay)).to(ctx.device), lens=torch.from_numpy(clamped)).cpu().numpy()
    got_over = ctx.encode_host(pay, lens=over, chunk_frames=9)
    assert np.array_equal(want_over.view(np.uint32), want_clamped.view(np.uint32)) and np.array_equal(got_over.view(np.uint32), want_clamped.view(np.uint32))
    wide_out = np.zeros((nfr, want_tx.shape[1] + 7), np.complex64)           # a strided destination takes the scatter path
    ctx.encode_host(pay, lens=lens, chunk_frames=7, out=wide_out)
    assert np.array_equal(wide_out[:, : want_tx.shape[1]].view(np.uint32), want_tx.view(np.uint32))
    # --- decode: those frames through the GPU channel, some slots empty
    tx = torch.from_numpy(want_tx).to(ctx.device)
    delay = torch.from_numpy(rng.integers(1, 65, nfr).astype(np.int32))
    fdl = torch.from_numpy((rng.random(nfr) * 1.8 - 0.9) * np.pi / 80)
    caps_d = ctx.channel_batch(tx, snr_db=30.0, seed=9, delay=delay, f_delta=fdl, span=span)
    caps_d[7].zero_(); caps_d[nfr - 1].zero_()
    D = ctx.data_symbols(nbytes)
    want = {k: v.cpu().numpy() for k, v in ctx.decode_batch(caps_d, max_symbols=D).items()}
    caps = alloc((nfr, span), np.complex64)
    caps[:] = caps_d.cpu().numpy()
    for chunk in (4, 16, nfr, 0):
        got = ctx.decode_host(caps, max_symbols=D, chunk_frames=chunk)
        for k in ("status", "len", "offset"):
            assert np.array_equal(got[k], want[k]), ("decode", chunk, k)
        assert np.array_equal(got["f_delta"], want["f_delta"]) and np.array_equal(got["metric"], want["metric"])
        for f in range(nfr):
            assert bytes(got["bytes"][f, : got["len"][f]]) == bytes(want["bytes"][f, : want["len"][f]]), ("decode", chunk, f)
        assert (got["status"][[7, nfr - 1]] == api.FRAME_NOSYNC).all() and (got["status"] == 0).sum() >= nfr - 4
    short = ctx.decode_host(caps[:, : 700], max_symbols=D)                   # frame_len < 800 after trimming: the reference's Err
    assert set(short["status"]) <= {api.FRAME_SHORT, api.FRAME_NOSYNC}
    # --- rx_demod (the BASELINE metric's path): regular streams of 16 symbols
    sym = alloc((nfr, 16 * 80), np.complex64)
    sym[:] = caps_d.cpu().numpy()[:, 800:800 + 1280]
    want_b = ctx.rx_demod(torch.from_numpy(np.ascontiguousarray(sym)).to(ctx.device), 16).cpu().numpy()
    for chunk in (3, 20, 0):
        out = alloc(want_b.shape, np.uint8)
        assert np.array_equal(ctx.demod_host(sym, 16, chunk_frames=chunk, out=out), want_b), ("demod", chunk)
    # other transform length + Hamming through the same pipe (row stride of the device block differs from the caller's)
    c2 = api.Context(n_fft=256, modulation=api.QAM16, guard_bands=False, ecc=api.ECC_HAMMING74)
    p2 = rng.integers(0, 256, (9, 333), dtype=np.uint8)
    t2 = c2.encode_host(p2, chunk_frames=2)
    assert rel_err(t2[4], orc.encode(orc.hamming74_encode(bytes(p2[4])), False, orc.QAM16, 256)) <= 1e-5
    r2 = c2.decode_host(np.concatenate([np.zeros((9, 40), np.complex64), t2, np.zeros((9, 400), np.complex64)], axis=1),
                        max_symbols=c2.data_symbols(333), chunk_frames=4)
    assert (r2["status"] == 0).all() and all(bytes(r2["bytes"][f, :333]) == bytes(p2[f]) for f in range(9))


def test_host_entry_points_reject_bad_arguments(api, orc):
    ctx = api.Context(n_fft=64, modulation=api.QPSK)
    x = np.zeros((2, 1000), np.complex64)
    out = np.zeros((2, 8), np.uint8)
    ln = np.zeros(2, np.int32)
    st = np.zeros(2, np.int32)
    lib, h = ctx.lib, ctx.h
    # rows of a host batch must not overlap; the output row must hold max_symbols of payload; NULL mandatory pointers
    assert lib.ofdm_rx_decode_host(h, x.ctypes.data, 2, 500, 1000, 0, 4, out.ctypes.data, 8, ln.ctypes.data, st.ctypes.data, None, None, None, 0) == -1
    assert lib.ofdm_rx_decode_host(h, x.ctypes.data, 2, 1000, 1000, 0, 40, out.ctypes.data, 8, ln.ctypes.data, st.ctypes.data, None, None, None, 0) == -1
    assert lib.ofdm_rx_decode_host(h, None, 2, 1000, 1000, 0, 1, out.ctypes.data, 8, ln.ctypes.data, st.ctypes.data, None, None, None, 0) == -1
    assert lib.ofdm_rx_decode_host(h, x.ctypes.data, 0, 1000, 1000, 0, 1, out.ctypes.data, 8, ln.ctypes.data, st.ctypes.data, None, None, None, 0) == 0
    assert lib.ofdm_tx_encode_host(h, out.ctypes.data, 2, 8, None, 8, x.ctypes.data, 100, 0) == -1   # out_stride < frame
    assert lib.ofdm_host_register(None, 10) == -1 and lib.ofdm_host_free(None) == 0
    # registering caller-owned memory makes it DMA-able in place, unregistering returns it
    buf = np.zeros(1 << 16, np.complex64)
    assert lib.ofdm_host_is_pinned(buf.ctypes.data, buf.nbytes) == 0
    assert lib.ofdm_host_register(buf.ctypes.data, buf.nbytes) == 0
    assert lib.ofdm_host_is_pinned(buf.ctypes.data, buf.nbytes) == 1
    assert lib.ofdm_host_unregister(buf.ctypes.data) == 0


def test_own_stream_contexts_side_by_side(api, orc):
    """ofdm_use_own_stream: two contexts with their own non-blocking streams give the results of the default-stream context."""
    import torch

    rng = np.random.default_rng(3)
    a = api.Context(n_fft=64, modulation=api.QAM64, guard_bands=True)
    b = api.Context(n_fft=64, modulation=api.QAM64, guard_bands=True)
    b.use_own_stream()
    pay = rng.integers(0, 256, (40, 560), dtype=np.uint8)
    ta = a.encode_host(pay)
    tb = b.encode_host(pay, chunk_frames=7)
    assert np.array_equal(ta.view(np.uint32), tb.view(np.uint32))
    x = torch.from_numpy(ta).to(a.device)
    torch.cuda.synchronize()
    rb = b.decode_batch(x, max_symbols=16)
    b.synchronize()                                   # b's stream is not torch's: results are ready only after its own sync
    ra = a.decode_batch(x, max_symbols=16)
    a.synchronize()
    assert bool((ra["bytes"] == rb["bytes"]).all()) and bool((ra["len"] == rb["len"]).all())
