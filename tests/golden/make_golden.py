"""Generate tests/golden/ofdm_golden.npz and tests/golden/reference_kats.json.

The reference is Rust (nightly, un-vendored git/path dependencies: SURVEY.md 8c) and cannot be built or run here, so
these vectors do NOT come from running it:
  * reference_kats.json  -- the known-answer DATA the reference's own tests / comments hold (inputs and expected
                            outputs, with the file:line they come from).  These are what pins the oracle.
  * ofdm_golden.npz      -- seeded inputs and the CPU oracle's outputs for every stage of the hot path (regression
                            vectors: "parity unpinned" beyond the KATs).  tests/test_golden.py checks that the oracle
                            still reproduces them (CPU) and that the HIP path matches them through the C ABI (GPU).
Run from the repo root:  python tests/golden/make_golden.py
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))
from oracle import oracle as orc  # noqa: E402
from util import fc32, through_channel, wide  # noqa: E402


def kats():
    return {
        "_comment": "known-answer data held by the reference's own tests and comments (SURVEY.md 8c (i)-(vi)); data only",
        "qpsk_roundtrip": {"src": "src/lib.rs:37-51", "payload_ascii": "alskdjas"},
        "qpsk_byte_0x61": {"src": "src/transmitter.rs:122-133", "byte": 0x61, "symbols_re_im": [[1, -1], [-1, -1], [-1, 1], [1, -1]]},
        "mean": {"src": "src/signals/mod.rs:385-394", "in_re_im": [[1, 1], [1, 2], [1, 3]], "out_re_im": [1, 2]},
        "xcorr": [
            {"src": "src/signals/mod.rs:420-430", "a": [1, 2, 3], "b": [4, 5], "full": [0, 5, 14, 23, 12], "idx_max": 3},
            {"src": "src/signals/mod.rs:431-441", "a": [1, 1, 0, 0, 1, 1, 0, 0], "b": [1, 1, 0, 0],
             "full": [0, 0, 0, 0, 0, 0, 1, 2, 1, 0, 1, 2, 1, 0, 0], "idx_max": 7},
        ],
        "bits": {"src": "src/utils.rs:281-327", "to_bools": {"255": [1] * 8, "0": [0] * 8, "127": [1] * 7 + [0]}},
        "angle": {"src": "src/receiver.rs:253-256", "z_re_im": [1, -1], "angle": -0.7853981633974483},
        "locking_signal": {"src": "src/transmitter.rs:60-72", "len": 80, "first": 0.375, "pre_shift_min": 0.25, "pre_shift_max": 0.496875},
        "frame_lengths": [
            {"src": "examples/lab3a.rs:11-46", "modulation": "qpsk", "guard": False, "payload_bytes": 400, "data_symbols": 26, "samples": 2880},
            {"src": "examples/lab3c.rs:15-54", "modulation": "bpsk", "guard": True, "payload_bytes": 765, "data_symbols": 131, "samples": 11280},
        ],
        "channel_taps_8_18": {"src": "src/channel.rs:26-31", "taps": [-0.1912, 0.9316, 0.2821, -0.1990, 0.1630, -0.1017, 0.0544, -0.0261, 0.0090, 0.0, -0.0034]},
    }


def golden():
    rng = np.random.default_rng(0x0FD3)
    g = {}
    # a8: FFT / IFFT
    x = fc32(rng.standard_normal(256) + 1j * rng.standard_normal(256))
    g["fft_in"] = x
    g["fft64_out"] = np.stack([orc.fft(wide(x[i * 64:(i + 1) * 64])) for i in range(4)])
    g["ifft256_out"] = orc.fft(wide(x), inverse=True)
    # a5 / EXT-1: constellations of every modulation, all byte values
    allb = bytes(range(256))
    for name, mod in (("bpsk", orc.BPSK), ("qpsk", orc.QPSK), ("qam16", orc.QAM16), ("qam64", orc.QAM64), ("qam256", orc.QAM256)):
        g[f"map_{name}"] = orc.modulate(allb, mod)
    g["map_bytes"] = np.frombuffer(allb, np.uint8)
    # EXT-2: Hamming(7,4)
    hin = bytes(rng.integers(0, 256, 32, dtype=np.uint8))
    g["ham_in"] = np.frombuffer(hin, np.uint8)
    g["ham_code"] = np.frombuffer(orc.hamming74_encode(hin), np.uint8)
    # a1: TX frames
    pay = bytes(rng.integers(0, 256, 150, dtype=np.uint8))
    g["payload"] = np.frombuffer(pay, np.uint8)
    g["tx_qpsk_noguard"] = orc.encode(pay, guard=False, modulation=orc.QPSK)
    g["tx_qam64_guard"] = orc.encode(pay, guard=True, modulation=orc.QAM64)
    g["tx_qam16_guard_n256"] = orc.encode(pay, guard=True, modulation=orc.QAM16, n_fft=256)
    # a17-a20: config-2 style RX demod (8 symbols, CP + FFT64 + demap, H == 1)
    nd = orc.data_carriers(64, True)
    dbytes = bytes(rng.integers(0, 256, 8 * nd * 6 // 8, dtype=np.uint8))
    pts = orc.modulate(dbytes, orc.QAM64)
    sy = np.concatenate([orc.prefix_block(orc.encode_block(pts[s * nd:(s + 1) * nd], 64, True)[0]) for s in range(8)])
    sy = fc32(sy + 0.002 * (rng.standard_normal(sy.size) + 1j * rng.standard_normal(sy.size)))
    g["demod_in"] = sy
    g["demod_bytes"] = np.frombuffer(orc.rx_demod(wide(sy), 64, True, orc.QAM64), np.uint8)
    g["demod_tx_bytes"] = np.frombuffer(dbytes, np.uint8)
    # EXT-3 + a10: config-3 style capture (delay, FIR channel, CFO, AWGN) -> timing, CFO, decoded bytes
    pay3 = bytes(rng.integers(0, 256, 560, dtype=np.uint8))
    tx3 = orc.encode(pay3, guard=True, modulation=orc.QAM64)
    cap = through_channel(orc, rng, tx3, 2176, 23, -0.0123, snr_db=30.0, data_start=800)
    g["cap_payload"] = np.frombuffer(pay3, np.uint8)
    g["cap"] = cap
    d_hat, p, m, fd = orc.sc_sync(wide(cap), 80, 3, 0, 0.5)
    g["cap_sc"] = np.array([d_hat, fd, m], np.float64)
    r = orc.decode_sc(wide(cap), guard=True, modulation=orc.QAM64)
    g["cap_decoded"] = np.frombuffer(bytes(r["bytes"]), np.uint8)
    g["cap_offset"] = np.array([r["offset"]], np.int64)
    return g


if __name__ == "__main__":
    json.dump(kats(), open(os.path.join(HERE, "reference_kats.json"), "w"), indent=1)
    g = golden()
    np.savez_compressed(os.path.join(HERE, "ofdm_golden.npz"), **g)
    print({k: (v.shape, str(v.dtype)) for k, v in g.items()})
