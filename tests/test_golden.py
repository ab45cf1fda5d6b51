"""Committed fixtures (tests/golden/, made by tests/golden/make_golden.py):
  * reference_kats.json -- the known-answer data of the reference's own tests / comments: pins the oracle;
  * ofdm_golden.npz     -- seeded inputs + oracle outputs per stage: the oracle must still reproduce them (CPU), and the
                           HIP path must match them through the C ABI (GPU): bytes / indices / timing bit-exact,
                           complex samples within 1e-5 norm-relative (BASELINE.json north_star).
"""
import json
import os

import numpy as np
import pytest

from util import rel_err, wide

HERE = os.path.dirname(os.path.abspath(__file__))
TOL = 1e-5


@pytest.fixture(scope="module")
def kats():
    return json.load(open(os.path.join(HERE, "golden", "reference_kats.json")))


@pytest.fixture(scope="module")
def gold():
    return np.load(os.path.join(HERE, "golden", "ofdm_golden.npz"), allow_pickle=False)


# ------------------------------------------------------------------ CPU: oracle vs the reference's own KAT data
def test_oracle_vs_reference_kats(orc, kats):
    k = kats["qpsk_roundtrip"]["payload_ascii"].encode()
    assert orc.demodulate(orc.modulate(k, orc.QPSK), orc.QPSK) == k
    k = kats["qpsk_byte_0x61"]
    assert [[z.real, z.imag] for z in orc.modulate(bytes([k["byte"]]), orc.QPSK)] == k["symbols_re_im"]
    k = kats["mean"]
    assert orc.mean([complex(*z) for z in k["in_re_im"]]) == complex(*k["out_re_im"])
    for k in kats["xcorr"]:
        idx, full = orc.xcorr_fft(k["a"], [float(v) for v in k["b"]])
        assert idx == k["idx_max"]
        np.testing.assert_allclose(full, k["full"], atol=1e-12)
    for b, bits in kats["bits"]["to_bools"].items():
        assert orc.to_bools(int(b)) == [bool(v) for v in bits]
    k = kats["angle"]
    assert abs(orc.angle(complex(*k["z_re_im"])) - k["angle"]) < 1e-15
    k = kats["locking_signal"]
    assert orc.locking_signal(k["len"])[0] == k["first"]
    for k in kats["frame_lengths"]:
        mod = {"bpsk": orc.BPSK, "qpsk": orc.QPSK}[k["modulation"]]
        assert orc.frame_len(k["payload_bytes"], 64, k["guard"], mod) == k["samples"] == 800 + 80 * k["data_symbols"]
    np.testing.assert_array_equal(orc.channel_taps()[8:19], kats["channel_taps_8_18"]["taps"])


# ------------------------------------------------------------------ CPU: oracle still reproduces the committed vectors
def test_oracle_reproduces_golden(orc, gold):
    x = wide(gold["fft_in"])
    for i in range(4):
        assert rel_err(orc.fft(x[i * 64:(i + 1) * 64]), gold["fft64_out"][i]) < 1e-13
    assert rel_err(orc.fft(x, inverse=True), gold["ifft256_out"]) < 1e-13
    allb = bytes(gold["map_bytes"])
    for name, mod in (("bpsk", orc.BPSK), ("qpsk", orc.QPSK), ("qam16", orc.QAM16), ("qam64", orc.QAM64), ("qam256", orc.QAM256)):
        np.testing.assert_array_equal(orc.modulate(allb, mod), gold[f"map_{name}"])
        assert orc.demodulate(gold[f"map_{name}"][: (len(gold[f"map_{name}"]) // 8) * 8], mod) == allb[: (len(gold[f"map_{name}"]) // 8) * 8 * mod // 8]
    assert orc.hamming74_encode(bytes(gold["ham_in"])) == bytes(gold["ham_code"])
    assert orc.hamming74_decode(bytes(gold["ham_code"]))[0] == bytes(gold["ham_in"])
    pay = bytes(gold["payload"])
    assert rel_err(orc.encode(pay, guard=False, modulation=orc.QPSK), gold["tx_qpsk_noguard"]) < 1e-13
    assert rel_err(orc.encode(pay, guard=True, modulation=orc.QAM64), gold["tx_qam64_guard"]) < 1e-13
    assert rel_err(orc.encode(pay, guard=True, modulation=orc.QAM16, n_fft=256), gold["tx_qam16_guard_n256"]) < 1e-13
    assert orc.rx_demod(wide(gold["demod_in"]), 64, True, orc.QAM64) == bytes(gold["demod_bytes"]) == bytes(gold["demod_tx_bytes"])
    d_hat, _, m, fd = orc.sc_sync(wide(gold["cap"]), 80, 3, 0, 0.5)
    assert d_hat == int(gold["cap_sc"][0]) and abs(fd - gold["cap_sc"][1]) < 1e-12 and abs(m - gold["cap_sc"][2]) < 1e-12
    r = orc.decode_sc(wide(gold["cap"]), guard=True, modulation=orc.QAM64)
    assert r["status"] == 0 and r["offset"] == int(gold["cap_offset"][0])
    assert bytes(r["bytes"]) == bytes(gold["cap_decoded"]) == bytes(gold["cap_payload"])


# ------------------------------------------------------------------ GPU: the HIP path against the committed vectors
@pytest.mark.gpu
def test_hip_matches_golden(ofdm, gold):
    import torch
    from ofdm_amd import api

    ctx = api.Context(n_fft=64, modulation=api.QAM64, guard_bands=True)
    x = ctx.to_device(gold["fft_in"])
    assert rel_err(ctx.fft(x.view(4, 64)).cpu().numpy(), gold["fft64_out"]) < TOL
    c256 = api.Context(n_fft=256, modulation=api.QAM16, guard_bands=True)
    assert rel_err(c256.fft(c256.to_device(gold["fft_in"]).view(1, 256), inverse=True).cpu().numpy()[0], gold["ifft256_out"]) < TOL
    allb = torch.from_numpy(gold["map_bytes"].copy()).to(ctx.device)
    for name, mod in (("bpsk", api.BPSK), ("qpsk", api.QPSK), ("qam16", api.QAM16), ("qam64", api.QAM64), ("qam256", api.QAM256)):
        c = api.Context(n_fft=64, modulation=mod, guard_bands=True)
        want = gold[f"map_{name}"]
        nb = (len(want) // 8) * 8 * mod // 8
        got = c.modulate(allb[:nb]).cpu().numpy()
        np.testing.assert_array_equal(got.astype(np.complex128), want[: nb * 8 // mod].astype(np.complex64).astype(np.complex128))
        back = c.demodulate(c.to_device(want[: nb * 8 // mod].astype(np.complex64)))
        assert bytes(back.cpu().numpy()) == bytes(gold["map_bytes"][:nb])
    code = ctx.hamming74_encode(torch.from_numpy(gold["ham_in"].copy()).to(ctx.device))
    assert bytes(code.cpu().numpy()) == bytes(gold["ham_code"])
    dec = ctx.hamming74_decode(torch.from_numpy(gold["ham_code"].copy()).to(ctx.device))
    assert bytes(dec[0].cpu().numpy()) == bytes(gold["ham_in"])
    pay = bytes(gold["payload"])
    assert rel_err(api.encode(pay, False, api.QPSK), gold["tx_qpsk_noguard"]) < TOL
    assert rel_err(api.encode(pay, True, api.QAM64), gold["tx_qam64_guard"]) < TOL
    assert rel_err(api.encode(pay, True, api.QAM16, n_fft=256), gold["tx_qam16_guard_n256"]) < TOL
    out = ctx.rx_demod(ctx.to_device(gold["demod_in"]).view(1, -1), syms_per_frame=8)
    assert bytes(out.cpu().numpy()[0]) == bytes(gold["demod_bytes"])
    cap = ctx.to_device(gold["cap"]).view(1, -1)
    d_hat, f_delta, metric = ctx.sc_correlate(cap)
    assert int(d_hat[0]) == int(gold["cap_sc"][0])
    assert abs(float(f_delta[0]) - gold["cap_sc"][1]) < 1e-9 and abs(float(metric[0]) - gold["cap_sc"][2]) < 1e-6
    r = ctx.decode_batch(cap, max_symbols=ctx.data_symbols(560))
    assert int(r["status"][0]) == 0 and int(r["offset"][0]) == int(gold["cap_offset"][0]) and int(r["len"][0]) == 560
    assert bytes(r["bytes"][0, :560].cpu().numpy()) == bytes(gold["cap_decoded"])
    assert api.decode(wide(gold["cap"]), True, api.QAM64) == bytes(gold["cap_payload"])
