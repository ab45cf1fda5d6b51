"""Pin the CPU oracle against every known-answer value the reference's own tests and comments hold
 and cross-check its FFT/convolution against numpy as an independent second opinion.

numpy is NOT the reference; it only confirms the oracle restates the textbook DFT contract of rustfft.
"""
import numpy as np
import pytest

CORPUS = (  # src/utils.rs:71-86 payload text is only used as payload bytes; any bytes do
    b"\nI met a traveller from an antique land,\nWho said-Two vast and trunkless legs of stone\n"
    b"Stand in the desert. . . . Near them, on the sand,\nHalf sunk a shattered visage lies, whose frown,\n"
)


def cycle_bytes(n):
    return bytes(CORPUS[i % len(CORPUS)] for i in range(n))


# ---- (i) src/lib.rs:37-51
def test_qpsk_roundtrip_lib_rs(orc):
    assert orc.demodulate(orc.modulate(b"alskdjas", orc.QPSK), orc.QPSK) == b"alskdjas"


# ---- SURVEY a5 KAT: byte 0x61 -> QPSK [(1,-1),(-1,-1),(-1,1),(1,-1)] (transmitter.rs:122-133, LSB first)
def test_qpsk_byte_0x61(orc):
    got = orc.modulate(b"\x61", orc.QPSK)
    assert list(got) == [1 - 1j, -1 - 1j, -1 + 1j, 1 - 1j]


def test_bpsk_levels(orc):
    got = orc.modulate(b"\x05", orc.BPSK)  # bits LSB first: 1,0,1,0,0,0,0,0
    assert list(got) == [1, -1, 1, -1, -1, -1, -1, -1]


# ---- (ii) src/signals/mod.rs:385-394
def test_mean_exact(orc):
    assert orc.mean([1 + 1j, 1 + 2j, 1 + 3j]) == 1 + 2j


# ---- (iii) src/signals/mod.rs:420-441 (full-length outputs; the test's "expected" arrays are sub-slices)
def test_xcorr_fft_kat(orc):
    idx, lags = orc.xcorr_fft([1, 2, 3], [4.0, 5.0])
    assert idx == 3
    np.testing.assert_allclose(lags, [0, 5, 14, 23, 12], atol=1e-12)
    np.testing.assert_allclose(lags[2:5], [14, 23, 12], atol=1e-12)  # the reference's expected slice
    idx, lags = orc.xcorr_fft([1, 1, 0, 0, 1, 1, 0, 0], [1, 1, 0, 0])
    assert idx == 7
    np.testing.assert_allclose(lags, [0, 0, 0, 0, 0, 0, 1, 2, 1, 0, 1, 2, 1, 0, 0], atol=1e-12)
    np.testing.assert_allclose(lags[7:15], [2, 1, 0, 1, 2, 1, 0, 0], atol=1e-12)


# ---- (iv) src/utils.rs:281-327
def test_get_bit_at(orc):
    assert orc.to_bools(255) == [True] * 8
    assert orc.to_bools(0) == [False] * 8
    assert orc.to_bools(127) == [True] * 7 + [False]


def test_bools_and_back(orc):
    for n in range(256):
        assert orc.bools_to_u8(orc.to_bools(n)) == n


def test_analysis_counts(orc):
    # counts from errs_is_right; rate per the formula at utils.rs:61 (the rates asserted there are inconsistent)
    assert orc.analysis(bytes([1, 0, 1, 0]), bytes([1, 0, 1, 0])) == (0, 0, 0.0)
    assert orc.analysis(bytes([1, 0, 0, 0]), bytes([1, 0, 1, 0])) == (1, 1, 1 / 32)
    assert orc.analysis(bytes([0, 0, 0, 0]), bytes([1, 0, 1, 0])) == (2, 2, 2 / 32)


# ---- (v) comment KATs: src/receiver.rs:253-256, src/channel.rs:99-177
def test_angle_comment(orc):
    assert abs(orc.angle(1 - 1j) - (-0.7854)) < 1e-4


MATLAB_HEAD = [0, 0, 0, 0, 0, 0, 0, 0, -0.1912, 0.7404, 1.0225, 0.8234, 0.9864, 0.8847, 0.9391, 0.9130, 0.9220,
               0.9220, 0.9186, 0.9186, 0.9186, 0.9186, 0.9186, 0.9186]
MATLAB_TAIL = [1.1098, 0.1782, -0.1039, 0.0952, -0.0678, 0.0339, -0.0205, 0.0056, -0.0034, -0.0034, 0.0]


def test_convolve_matlab_listing(orc):
    # channel.rs:99-177 lists conv((1-1i)*ones(16), CHANNEL) to 4 decimals (MATLAB used unrounded taps: 2e-4)
    out = orc.convolve((1 - 1j) * np.ones(16), orc.channel_taps())
    np.testing.assert_allclose(out.real[:24], MATLAB_HEAD, atol=2.5e-4)
    np.testing.assert_allclose(out.imag[:24], -np.array(MATLAB_HEAD), atol=2.5e-4)
    np.testing.assert_allclose(out.real[24:35], MATLAB_TAIL, atol=2.5e-4)
    assert np.abs(out[35:]).max() < 1e-12


# ---- (vi) closed forms: transmitter.rs:63-69 and frame lengths
def test_locking_signal_closed_form(orc):
    lock = orc.locking_signal(80)
    assert lock[0] == 0.375
    assert lock.real.min() == 0.25 and lock.real.max() == 0.496875
    assert np.all(lock.imag == 0)
    # pre-shift ramp is monotone; fft_shift of an even length swaps halves
    np.testing.assert_array_equal(lock[:40].real, 0.5 * (np.arange(40, 80) / 160 + 0.5))
    np.testing.assert_array_equal(lock[40:].real, 0.5 * (np.arange(0, 40) / 160 + 0.5))


def test_frame_lengths(orc):
    assert orc.frame_len(400, 64, False, orc.QPSK) == 800 + 80 * 26 == 2880  # examples/lab3a.rs
    assert orc.frame_len(765, 64, True, orc.BPSK) == 800 + 80 * 131          # examples/lab3c.rs (500 B + RS)
    assert orc.frame_len(560, 64, True, orc.QAM64) == 2080                    # BASELINE config 1
    assert orc.encode(cycle_bytes(400), False, orc.QPSK).size == 2880


# ---- numpy cross-checks (independent implementation of the same mathematics)
@pytest.mark.parametrize("n", [1, 2, 5, 64, 77, 80, 159, 1024, 4096, 5885])
def test_fft_matches_numpy(orc, n):
    rng = np.random.default_rng(n)
    x = rng.standard_normal(n) + 1j * rng.standard_normal(n)
    scale = np.linalg.norm(np.fft.fft(x))
    assert np.linalg.norm(orc.fft(x) - np.fft.fft(x)) <= 1e-12 * scale
    assert np.linalg.norm(orc.fft(x, inverse=True) - np.fft.ifft(x)) <= 1e-12 * np.linalg.norm(x)


def test_shifts_match_numpy(orc):
    for n in (1, 2, 7, 8, 80):
        x = np.arange(n) + 0j
        np.testing.assert_array_equal(orc.fft_shift(x), np.fft.fftshift(x))
        np.testing.assert_array_equal(orc.ifft_shift(x), np.fft.ifftshift(x))


def test_convolve_matches_numpy(orc):
    rng = np.random.default_rng(3)
    a = rng.standard_normal(300) + 1j * rng.standard_normal(300)
    np.testing.assert_allclose(orc.convolve(a, orc.channel_taps()), np.convolve(a, orc.channel_taps()), atol=1e-12)


def test_variance_is_pseudo_variance(orc):
    x = np.array([1 + 1j, 2 - 1j, 0.5 + 3j])
    m = x.mean()
    assert abs(orc.variance(x) - ((m - x) ** 2).sum() / 3) < 1e-15


def test_fc32_roundtrip(orc):
    x = np.array([0.5 - 0.25j, 1e-3 + 3j])
    f = orc.sig_to_fc32(x)
    assert f.dtype == np.float32 and list(f) == [0.5, -0.25, np.float32(1e-3), 3.0]
    np.testing.assert_array_equal(orc.fc32_to_sig(f), f[0::2].astype(np.float64) + 1j * f[1::2].astype(np.float64))


# ---- carrier map (transmitter.rs:150-161) and its k-fold tiling (EXT-4)
def test_carrier_map(orc):
    cls = [orc.carrier_class(i, 64, True) for i in range(64)]
    nulls = [i for i, c in enumerate(cls) if c == 1]
    pilots = [i for i, c in enumerate(cls) if c == 2]
    assert nulls == [0, 1, 2, 3, 4, 5, 32, 59, 60, 61, 62, 63]
    assert pilots == [6, 25, 39, 58]
    assert orc.data_carriers(64, True) == 48 and orc.data_carriers(64, False) == 64
    assert [orc.carrier_class(i, 64, False) for i in range(64)] == [0] * 64
    cls1024 = [orc.carrier_class(i, 1024, True) for i in range(1024)]
    assert cls1024 == [c for c in cls for _ in range(16)]
    assert orc.data_carriers(4096, True) == 48 * 64


def test_encode_block_guard(orc):
    # transmitter.rs bands_work: 52 symbols offered, 48 consumed, nulls 0, pilots 1
    out, used = orc.encode_block(np.arange(1, 53) + 0j, 64, True)
    assert used == 48
    assert out[6] == 1 and out[25] == 1 and out[39] == 1 and out[58] == 1
    assert np.all(out[[0, 1, 2, 3, 4, 5, 32, 59, 60, 61, 62, 63]] == 0)
    data_bins = [i for i in range(64) if orc.carrier_class(i, 64, True) == 0]
    np.testing.assert_array_equal(out[data_bins], np.arange(1, 49))
    out, used = orc.encode_block(np.arange(1, 11) + 0j, 64, False)  # stream runs dry -> zeros
    assert used == 10 and np.all(out[10:] == 0)


def test_prefix_block_cyclic(orc):
    x = np.arange(1, 9) + 0j  # transmitter.rs cyclic_prefix_works shape (ifft then last PREFIX + all)
    out = orc.prefix_block(x, cp=3)
    t = np.fft.ifft(x)
    np.testing.assert_allclose(out, np.concatenate([t[-3:], t]), atol=1e-15)


def test_normalize_signed_max(orc):
    x = np.array([-3 + 0.5j, 0.25 - 2j])
    np.testing.assert_array_equal(orc.normalize(x), x / 0.5)  # max over +re/+im only (transmitter.rs:183-194)


# ---- QPSK tie rules Q6 (receiver.rs:169-175)
def test_qpsk_tie_rules(orc):
    pts = np.array([0 + 0j, 1 + 0j, -1 + 0j, 0 - 1j, -1 + 1j, -1 - 1j, complex(-0.0, 0.0), complex(1, np.nan)])
    idx = orc.demap_indices(pts, orc.QPSK)
    # (l, r) -> l | r << 1
    assert list(idx) == [3, 3, 0, 1, 2, 0, 3, 0]


# ---- EXT-1 definition checks (parity unpinned by the reference: the oracle is the definition)
def test_qam64_gray_and_levels(orc):
    sym = orc.modulate(bytes(range(256)) * 3, orc.QAM64)
    lv = np.unique(np.round(sym.real * 7).astype(int))
    assert list(lv) == [-7, -5, -3, -1, 1, 3, 5, 7] and np.abs(sym.real).max() == 1.0
    # 802.11a-style Gray: first bit 1 => positive half; neighbours differ in one bit per axis
    idx = orc.demap_indices(sym, orc.QAM64)
    lvl = np.round((sym.real * 7 + 7) / 2).astype(int)
    for a in range(7):
        ia = idx[lvl == a][0] & 7
        ib = idx[lvl == a + 1][0] & 7
        assert bin(ia ^ ib).count("1") == 1
    assert all((i & 1) == 1 for i, s in zip(idx, sym) if s.real > 0)
    # ties resolve upward (>= boundary -> upper level)
    assert orc.demap_indices(np.array([0 + 0j]), orc.QAM64)[0] == orc.demap_indices(np.array([1 / 7 + 1j / 7]), orc.QAM64)[0]


@pytest.mark.parametrize("mod", [1, 2, 4, 6, 8])
def test_mod_demod_roundtrip(orc, mod):
    rng = np.random.default_rng(mod)
    data = bytes(rng.integers(0, 256, 96, dtype=np.uint8))
    sym = orc.modulate(data, mod)
    assert sym.size == 96 * 8 // mod
    assert orc.demodulate(sym, mod) == data
    noisy = sym + (rng.standard_normal(sym.size) + 1j * rng.standard_normal(sym.size)) * 0.02 / (2 ** (mod // 2))
    assert orc.demodulate(noisy, mod) == data


def test_demodulate_requires_multiple_of_8(orc):
    assert orc.demodulate(np.ones(7) + 0j, orc.BPSK) == b""  # assert_eq!(remainder.len(), 0), receiver.rs:153


# ---- EXT-2 Hamming(7,4)
def test_hamming74(orc):
    rng = np.random.default_rng(0)
    data = bytes(rng.integers(0, 256, 64, dtype=np.uint8))
    code = orc.hamming74_encode(data)
    assert len(code) == 112
    assert orc.hamming74_decode(code) == (data, 0)
    # one flipped bit in every 7-bit codeword is corrected
    bits = np.unpackbits(np.frombuffer(code, np.uint8), bitorder="little").copy()
    for cw in range(bits.size // 7):
        bits[cw * 7 + int(rng.integers(0, 7))] ^= 1
    dec, fixed = orc.hamming74_decode(np.packbits(bits, bitorder="little").tobytes())
    assert dec == data and fixed == bits.size // 7
    assert orc.hamming74_encode(b"abcde")[:7] == orc.hamming74_encode(b"abcd")
    assert len(orc.hamming74_encode(b"abcde")) == 14 and orc.hamming74_decode(orc.hamming74_encode(b"abcde"))[0] == b"abcde\0\0\0"


# ---- a14 / a15
def test_frequency_correction_and_rotate(orc):
    rng = np.random.default_rng(5)
    left = rng.standard_normal(80) + 1j * rng.standard_normal(80)
    fd = 0.01
    right = left * np.exp(1j * fd * 80)
    assert abs(orc.frequency_correction(left, right) - fd) < 1e-12
    assert abs(orc.frequency_correction(right, left) - fd) < 1e-12  # abs() discards the sign (Q2)
    x = np.exp(1j * fd * np.arange(10, 110))
    np.testing.assert_allclose(orc.cfo_rotate(x, fd, 10), np.ones(100), atol=1e-13)


# ---- EXT-3 Schmidl-Cox on a clean repeated preamble: single peak exactly at the preamble start
def test_sc_sync_clean(orc):
    rng = np.random.default_rng(9)
    pre = orc.default_preamble(80)
    r = np.concatenate([0.01 * (rng.standard_normal(37) + 1j * rng.standard_normal(37)), orc.locking_signal(80),
                        np.tile(pre, 4), 0.3 * (rng.standard_normal(400) + 1j * rng.standard_normal(400))])
    fd = -0.02
    r = r * np.exp(1j * fd * np.arange(r.size))
    d, P, M, fdh = orc.sc_sync(r, 80, 3, 0)
    assert d == 37 + 80 and abs(M - 1) < 1e-12 and abs(fdh - fd) < 1e-12
    m, p = orc.sc_metric(r, 80, 3, 0)
    assert int(np.argmax(m)) == d and m.size == r.size - 320 + 1
    # all-zero input: no valid lag
    assert orc.sc_sync(np.zeros(1000) + 0j, 80, 3, 0)[0] == -1


# ---- loop-back: examples/lab3a.rs:11-46 and lab3b.rs:12-38 wiring (QPSK, no guard, 400 B, 30 dB)
@pytest.mark.parametrize("timing_error", [False, True])
def test_loopback_lab3(orc, timing_error):
    data = cycle_bytes(400)
    tx = orc.encode(data, False, orc.QPSK)
    rx, fd = orc.channel(tx, 30.0, timing_error, seed=11)
    assert rx.size == tx.size + 63
    res = orc.decode_ref(rx, False, orc.QPSK)
    assert res["status"] == 0 and res["offset"] == 8  # lag 9 (main tap) - 1 (Q1)
    if timing_error:
        assert abs(res["f_delta"] - fd) < 2e-4
    assert len(res["bytes"]) == 400
    assert orc.analysis(data, res["bytes"])[0] == 0


# ---- examples/lab3c_image.rs: the reference's one data fixture (support/dancing.bytes, a 24 x 24 palette frame: DATA, copied to
#      tests/golden/dancing.bytes) -> create_transmission_bytes -> encode!(guard_bands) [BPSK] -> channel -> decode! ->
#      decipher_transmission_bytes (src/utils.rs:97-180)
def dancing_bytes():
    import os
    return open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "dancing.bytes"), "rb").read()


def test_lab3c_image_fixture_over_the_link(orc):
    img = dancing_bytes()
    assert len(img) == 576 == 24 * 24                       # examples/lab3c_image.rs:70 `let dims = (24, 24)`
    coded = orc.create_transmission_bytes(img)
    assert len(coded) == 3 * 255                            # 576 = 223 + 223 + 130 -> three RS(255,223) blocks (zero-filled tail)
    tx = orc.encode(coded, True, orc.BPSK)                  # ofdm::encode!(data, guard_bands): modulation defaults to BPSK
    assert tx.size == 800 + 80 * 131                        # 16 + 765 bytes at 48 bits per symbol
    rx, _ = orc.channel(tx, 30.0, True, seed=24)
    for res in (orc.decode_ref(rx, True, orc.BPSK), orc.decode_sc(rx, True, orc.BPSK, cfo_abs=True)):
        assert res["status"] == 0 and bytes(res["bytes"]) == coded
        plain = orc.decipher_transmission_bytes(bytes(res["bytes"]))
        # (765 = 3 x 255 exactly: the reference's loop then decodes one more, all-zero block at end of input, utils.rs:172-176)
        assert plain is not None and len(plain) == 4 * 223 and plain[:576] == img and not any(plain[576:])
    # sixteen byte errors in one block are repaired, seventeen are not (t = 16)
    bad = bytearray(coded)
    for i in range(16):
        bad[255 + 7 * i] ^= 0x5A
    assert orc.decipher_transmission_bytes(bytes(bad))[:576] == img
    bad[255 + 7 * 16] ^= 0x5A
    assert orc.decipher_transmission_bytes(bytes(bad)) is None


def test_decode_too_short(orc):
    res = orc.decode_ref(np.concatenate([np.zeros(5) + 0j, orc.locking_signal(80), np.zeros(300) + 0j]), False, orc.BPSK)
    assert res["status"] == -1  # "Input not long enough, bailing early" (receiver.rs:27-29)


# ---- BASELINE config 1: one 64-carrier 64-QAM frame, encode -> channel(30 dB, CFO) -> decode on CPU
def test_config1_loopback_64qam(orc):
    data = cycle_bytes(560)
    tx = orc.encode(data, True, orc.QAM64)
    assert tx.size == 2080
    rx, fd = orc.channel(tx, 30.0, True, seed=1)
    ref = orc.decode_ref(rx, True, orc.QAM64)
    sc = orc.decode_sc(rx, True, orc.QAM64, backoff=4, cfo_abs=True)
    for res in (ref, sc):
        assert res["status"] == 0 and len(res["bytes"]) == 560
        ne, nb, rate = orc.analysis(data, res["bytes"])
        assert rate <= 1e-3
    assert abs(sc["f_delta"] - fd) < 1e-4 and sc["offset"] == 9 + 80 - 80 - 4
    # an all-zero payload makes every data symbol identical (period-80 like the preamble): the
    # "first threshold crossing, then peak" rule still locks onto the preamble
    tx0 = orc.encode(bytes(560), True, orc.QAM64)
    rx0, _ = orc.channel(tx0, 30.0, True, seed=3)
    sc0 = orc.decode_sc(rx0, True, orc.QAM64, cfo_abs=True)
    assert sc0["status"] == 0 and sc0["offset"] == 5 and sc0["bytes"] == bytes(560)
