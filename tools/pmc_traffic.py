"""profiles/rNN_pmc_traffic.json from the FETCH_SIZE / WRITE_SIZE summary of tools/profile_round.sh and the byte counts
tools/pmc_probe.py printed:  python tools/pmc_traffic.py <fetch_write_summary.tsv> <pmc_FETCH_SIZE.log> <round> > profiles/<round>_pmc_traffic.json
FETCH_SIZE / WRITE_SIZE are KiB per dispatch; on gfx950 FETCH_SIZE reports half of a coalesced stream
(/opt/skills/guides/MI355X_MICROARCH.md, HBM section) -- the factor is CALIBRATED here on a kernel whose traffic is known
(the batched FFT k_sym<64,0>: reads and writes n * 512 B) and applied to every kernel."""
import json, sys

tsv, log, rnd = sys.argv[1:4]
cnt, cmax = {}, {}
for line in open(tsv):
    name, ctr, calls, mean, mx = line.rstrip("\n").split("\t")[:5]
    cnt[(name.replace("void ", ""), ctr)] = float(mean.split("=")[1])
    cmax[(name.replace("void ", ""), ctr)] = float(mx.split("=")[1])
info = next(json.loads(l) for l in open(log) if l.startswith("{"))


def kib(kernel, ctr):
    hits = [v for (n, c), v in cnt.items() if c == ctr and n.startswith(kernel)]
    if len(hits) != 1:
        raise SystemExit(f"{kernel}/{ctr}: {len(hits)} matches")
    return hits[0]


fr = kib("ofdm::k_sym<64, 0>", "FETCH_SIZE")
fw = kib("ofdm::k_sym<64, 0>", "WRITE_SIZE")
rf = info["fft_bytes_each_way"] / (fr * 1024.0)      # read factor (about 2 on gfx950)
wf = info["fft_bytes_each_way"] / (fw * 1024.0)      # write factor (about 1)


def rd(kernel): return kib(kernel, "FETCH_SIZE") * 1024.0 * rf
def wr(kernel): return kib(kernel, "WRITE_SIZE") * 1024.0 * wf
def rd_max(kernel):   # the largest call (a kernel that also runs over near-empty frame lists)
    hits = [v for (n, c), v in cmax.items() if c == "FETCH_SIZE" and n.startswith(kernel)]
    if len(hits) != 1:
        raise SystemExit(f"{kernel}/FETCH_SIZE: {len(hits)} matches")
    return hits[0] * 1024.0 * rf


n = info["frames_cfg2_cfg3"]; n4 = info["frames_cfg4"]; n5 = info["symbols_cfg5"]
cap3 = info["cfg3_capture_bytes"] / n; cap4 = info["cfg4_capture_bytes"] / n4
out = {
    "source": f"rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, tools/profile_round.sh) on tools/pmc_probe.py (packet frames "
              f"through the library's TX and GPU channel model), MI355X, round {rnd}; raw means in profiles/{rnd}_pmc_fetch_write_summary.tsv; "
              f"this file is tools/pmc_traffic.py's output",
    "units": "bytes per frame / symbol; FETCH_SIZE and WRITE_SIZE are KiB per dispatch, multiplied by the factors calibrated below",
    "calibration": {"kernel": "ofdm::k_sym<64,0> (batched FFT, 8 B per lane)", "known_bytes_each_way": info["fft_bytes_each_way"],
                    "FETCH_SIZE_KiB": fr, "WRITE_SIZE_KiB": fw, "read_factor": round(rf, 4), "write_factor": round(wf, 4)},
    "k_demod64": {"frames": n, "FETCH_SIZE_KiB": kib("ofdm::k_demod64<6, true, false, 16>", "FETCH_SIZE"),
                  "WRITE_SIZE_KiB": kib("ofdm::k_demod64<6, true, false, 16>", "WRITE_SIZE"),
                  "read_bytes_per_frame": round(rd("ofdm::k_demod64<6, true, false, 16>") / n, 1),
                  "write_bytes_per_frame": round(wr("ofdm::k_demod64<6, true, false, 16>") / n, 1),
                  "algorithmic_read_bytes_per_frame": 10240, "algorithmic_write_bytes_per_frame": 576,
                  "note": "16 symbols x 512 B: the 128-byte cyclic prefix of every 640-byte symbol is never fetched"},
    "cfg3": {"frames": n, "capture_bytes_per_frame": cap3,
             "every_lag_k_sc_cf_256_read_bytes_per_frame": round(rd_max("ofdm::k_sc_cf<256, 2, 4, 0, false>") / n, 1),
             "staged_first_lags_k_sc_cf_128_read_bytes_per_frame": round(rd("ofdm::k_sc_cf<128, ") / n, 1),
             "staged_k_rxframe64_read_bytes_per_frame": round(rd("ofdm::k_rxframe64<6, true, 0>") / n, 1),
             "staged_k_rxframe64_needed_bytes_per_frame": 21 * 512,
             "one_pass_k_sc_cf_read_bytes_per_frame": round(rd("ofdm::k_sc_cf<256, 2, 3, 6, true>") / n, 1),
             "one_pass_write_bytes_per_frame": round(wr("ofdm::k_sc_cf<256, 2, 3, 6, true>") / n, 1),
             "k_txframe64_read_bytes_per_frame": round(rd("ofdm::k_txframe64<6, true>") / n, 1),
             "k_txframe64_write_bytes_per_frame": round(wr("ofdm::k_txframe64<6, true>") / n, 1),
             "note": "the product's search reads the first 576 + W + L samples of a slot (k_sc_cf<128>) and the whole slot only for frames those "
                     "lags do not determine; k_rxframe64 loads 8 B per lane at arbitrary sample offsets (FETCH_SIZE is not reliable for that "
                     "pattern: 6.8 KB in round 3, 14.9 KB in round 4 for the same 10752 B needed), so the staged chain is claimed at first-lags bytes + 10752 B per frame"},
    "cfg4": {"frames": n4, "capture_bytes_per_frame": cap4, "search": "every lag",
             "k_sc_stream_read_bytes_per_frame": round(rd("ofdm::k_sc_stream<2>") / n4, 1),
             "k_rxframe1024_read_bytes_per_frame": round(rd("ofdm::k_rxframe1024<6, true>") / n4, 1),
             "k_rxframe1024_needed_bytes_per_frame": 9 * 8192,
             "k_rxframe1024_write_bytes_per_frame": round(wr("ofdm::k_rxframe1024<6, true>") / n4, 1),
             "chain_read_over_capture": round((rd("ofdm::k_sc_stream<2>") + rd("ofdm::k_rxframe1024<6, true>")) / n4 / cap4, 3),
             "note": "k_sc_stream stops reading a slot once the peak window has closed (the packet sits at the start of its slot), "
                     "k_rxframe1024 reads the 9 symbols it transforms: the whole chain moves about one capture (round 2: 2.33 captures "
                     "through k_scb_chunks + k_scb_fine + k_rxframe1024)"},
    "cfg5": {"symbols": n5, "k_demod4096_read_bytes_per_symbol": round(rd("ofdm::k_demod4096<8, true, false>") / n5, 1),
             "k_demod4096_write_bytes_per_symbol": round(wr("ofdm::k_demod4096<8, true, false>") / n5, 1),
             "k_tx4096_read_bytes_per_symbol": round(rd("ofdm::k_tx4096<true>") / n5, 1),
             "k_tx4096_write_bytes_per_symbol": round(wr("ofdm::k_tx4096<true>") / n5, 1),
             "note": "RX: 4096 x 8 B, the 1024-sample cyclic prefix is never fetched; TX: exactly the 5120-sample symbol is written, the payload (3072 B) read once"},
    "mid_kernels": {},
    # flat per-kernel view (bytes per frame of the probe's batches) for the bench blocks' `roofline.traffic` (tools/bench_cfg3._traffic)
    "per_kernel": {
        "k_sc_cf_128_first_lags": {"read_bytes_per_frame": round(rd("ofdm::k_sc_cf<128, ") / n, 1), "write_bytes_per_frame": round(wr("ofdm::k_sc_cf<128, ") / n, 1)},
        "k_rxframe64": {"read_bytes_per_frame": round(rd("ofdm::k_rxframe64<6, true, 0>") / n, 1), "write_bytes_per_frame": round(wr("ofdm::k_rxframe64<6, true, 0>") / n, 1),
                        "note": "8-byte loads at arbitrary sample offsets: a 512-byte symbol that starts anywhere touches five 128-byte lines (640 B), so 21 symbols fetch up to 13.4 KB for the 10 752 B the kernel needs; round 3's three-waves-per-SIMD kernel showed 6.8 KB here (FETCH_SIZE is not reliable for this pattern)"},
        "k_sc_stream": {"read_bytes_per_frame": round(rd("ofdm::k_sc_stream<2>") / n4, 1), "write_bytes_per_frame": round(wr("ofdm::k_sc_stream<2>") / n4, 1)},
        "k_rxframe1024": {"read_bytes_per_frame": round(rd("ofdm::k_rxframe1024<6, true>") / n4, 1), "write_bytes_per_frame": round(wr("ofdm::k_rxframe1024<6, true>") / n4, 1)},
    },
}
for nn, r in ((512, 8), (2048, 32)):
    m = info[f"mid_{nn}"]
    s = m["symbols"]
    out["mid_kernels"][f"N{nn}"] = {
        "symbols": s, "symbol_bytes_with_prefix": (nn + nn // 4) * 8, "k_demod_mid_needed_bytes_per_symbol": nn * 8,
        "k_demod_mid_read_bytes_per_symbol": round(rd(f"ofdm::k_demod_mid<{r}, 6, true, false>") / s, 1),
        "k_demod_mid_write_bytes_per_symbol": round(wr(f"ofdm::k_demod_mid<{r}, 6, true, false>") / s, 1),
        "k_tx_mid_read_bytes_per_symbol": round(rd(f"ofdm::k_tx_mid<{r}, true>") / s, 1),
        "k_tx_mid_write_bytes_per_symbol": round(wr(f"ofdm::k_tx_mid<{r}, true>") / s, 1),
        "k_txframe_mid_frames": m["frames"],
        "k_txframe_mid_write_over_frame_bytes": round(wr(f"ofdm::k_txframe_mid<{r}, true, ") / m["frame_bytes"], 4)}
print(json.dumps(out, indent=1))
