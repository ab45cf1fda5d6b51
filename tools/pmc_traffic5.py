"""profiles/r05_pmc_traffic.json from the raw L2 memory-side counters of tools/pmc_round5.sh.
   python tools/pmc_traffic5.py gpurun_out/prof r05 > profiles/r05_pmc_traffic.json
Reads: bytes = TCC_EA0_RDREQ_sum x 128 B -- on gfx950 every read request the L2 sends to the fabric is one 128-byte line
(TCC_EA0_RDREQ_32B_sum and TCC_BUBBLE_sum are zero on every kernel here), which the derived FETCH_SIZE of ROCm 7.2 tallies at 64 B.
CALIBRATED on k_read_probe, whose byte counts are known exactly, in the frame kernels' own access pattern (8 B per lane, 512 B of every
640-byte symbol), over whole symbols, and with unit-stride 16-byte loads; the factor is then applied to every kernel.  The read group
was collected twice (rd_a, rd_b): the reproducibility of every figure is stated.  Writes: TCC_EA0_WRREQ_sum x 64 B, calibrated on the
batched FFT k_sym<64,0> (writes n x 512 B)."""
import json, os, sys

d, rnd = sys.argv[1], sys.argv[2]


def load(tag):
    t = {}
    for line in open(os.path.join(d, f"{rnd}_pmc5_{tag}.tsv")):
        f = line.rstrip("\n").split("\t")
        name = f[0].replace("void ", "")
        vals = [float(v) for v in f[5].split("=")[1].split(",")] if len(f) > 5 and f[5].startswith("vals=") else []
        t[(name, f[1])] = vals
    return t


info = None
for line in open(os.path.join(d, f"{rnd}_pmc5_rd_a.log")):
    if line.startswith("{"):
        info = json.loads(line)
rd_a, rd_b, l2, wrt = load("rd_a"), load("rd_b"), load("l2"), load("wr")
fetch, write = load("fetch"), load("write")
n, n4, n5 = info["frames_cfg2_cfg3"], info["frames_cfg4"], info["symbols_cfg5"]


def one(t, kernel, ctr):
    hits = [(k, v) for (k, c), v in t.items() if c == ctr and k.startswith(kernel)]
    if len(hits) != 1:
        raise SystemExit(f"{kernel}/{ctr}: {len(hits)} matches")
    return hits[0][1]


# ---- calibration: bytes per read request on the three probe patterns (dispatch order: pattern 0, 1, 2, twice)
probe = one(rd_a, "ofdm::k_read_probe", "TCC_EA0_RDREQ_sum")
known = [info["probe_symbols"] * b for b in (512, 640, 640)] * 2
per_req = [known[i] / probe[i] for i in range(len(probe))]
RD = 128.0
assert all(abs(p - RD) / RD < 0.005 for p in per_req), per_req
fftw = one(wrt, "ofdm::k_sym<64, 0>", "TCC_EA0_WRREQ_sum")
WR = 64.0
wr_per_req = [info["fft_bytes_each_way"] / v for v in fftw]


def rd(kernel, which=None, t=rd_a):
    v = one(t, kernel, "TCC_EA0_RDREQ_sum")
    v = v if which is None else [v[i] for i in which]
    return sum(v) / len(v) * RD


def wr(kernel, which=None):
    v = one(wrt, kernel, "TCC_EA0_WRREQ_sum")
    v = v if which is None else [v[i] for i in which]
    return sum(v) / len(v) * WR


def repro(kernel, which=None):
    a, b = rd(kernel, which, rd_a), rd(kernel, which, rd_b)
    return round(abs(a - b) / max(a, b), 6)


def kern(name, units, which=None, **extra):
    r = {"read_bytes_per_frame": round(rd(name, which) / units, 1), "write_bytes_per_frame": round(wr(name, which) / units, 1),
         "read_two_passes_relative_difference": repro(name, which)}
    try:
        h, m = one(l2, name, "TCC_HIT_sum"), one(l2, name, "TCC_MISS_sum")
        if which is not None:
            h, m = [h[i] for i in which], [m[i] for i in which]
        r["l2_hit_rate"] = round(sum(h) / (sum(h) + sum(m)), 4)
    except SystemExit:
        pass
    try:   # what the derived counter of rounds 1-4 says for the same launches (KiB -> bytes, its own 64-byte tally)
        fv = one(fetch, name, "FETCH_SIZE")
        fv = fv if which is None else [fv[i] for i in which]
        r["FETCH_SIZE_bytes_per_frame_uncorrected"] = round(sum(fv) / len(fv) * 1024.0 / units, 1)
    except SystemExit:
        pass
    r.update(extra)
    return r


SC80 = "ofdm::k_sc80<2>"
out = {
    "source": f"rocprofv3 --pmc, raw L2 memory-side counters, one group per run, nothing but --pmc (tools/pmc_round5.sh on tools/pmc_probe5.py), "
              f"MI355X, round {rnd}; per-dispatch values in profiles/{rnd}_pmc5_*.tsv; this file is tools/pmc_traffic5.py's output",
    "counters": "TCC_EA0_RDREQ_sum x 128 B, TCC_EA0_WRREQ_sum x 64 B",
    "units": "bytes per frame (per symbol for config 5)",
    "calibration": {
        "reads": {"kernel": "ofdm::k_read_probe, patterns 0 (k_demod64's: 8 B per lane, 512 B of every 640-byte symbol), 1 (whole symbols), 2 (16 B per lane, unit stride), each twice",
                  "known_bytes": known, "TCC_EA0_RDREQ_sum": probe, "bytes_per_request": [round(p, 3) for p in per_req], "factor_used": RD,
                  "TCC_EA0_RDREQ_32B_sum_and_TCC_BUBBLE_sum": "0 on every kernel of the workload"},
        "writes": {"kernel": "ofdm::k_sym<64,0> (batched FFT: writes n x 512 B)", "known_bytes": info["fft_bytes_each_way"],
                   "TCC_EA0_WRREQ_sum": fftw, "bytes_per_request": [round(p, 3) for p in wr_per_req], "factor_used": WR},
    },
    "k_demod64": kern("ofdm::k_demod64<6, true, false, 16>", n, algorithmic_read_bytes_per_frame=10240, algorithmic_write_bytes_per_frame=576,
                      counters="TCC_EA0_RDREQ x 128 B / TCC_EA0_WRREQ x 64 B",
                      note="16 symbols x 512 B: the 128-byte cyclic prefix of every 640-byte symbol is a whole aligned line and is never fetched"),
    "cfg3": {"frames": n, "slot_bytes_per_frame": info["cfg3_capture_bytes"] / n,
             "k_sc80_stated_placement": kern(SC80, n, [0, 1]),
             "k_sc80_late_packets_and_empty_slots": kern(SC80, n, [2, 4], slot_bytes_per_frame=info["late_capture_bytes"] / n),
             "k_sc80_noise_only_slots": kern(SC80, n, [3, 5], slot_bytes_per_frame=info["noise_capture_bytes"] / n),
             "k_rxframe64": kern("ofdm::k_rxframe64<6, true, 0>", n, needed_bytes_per_frame=21 * 512,
                                 note="8-byte loads at the frame's own (arbitrary) sample offset: the 128-byte lines between two symbols hold the end of one "
                                      "symbol, a cyclic prefix and the start of the next, so every line of the 21 x 640-byte region is fetched (13.4 KB) although "
                                      "only 21 x 512 B are used; the rest is the slot's first line (the branch-free loads' safe address) and the per-frame scalars"),
             "k_sc_cf_256_every_lag": kern("ofdm::k_sc_cf<256, 2, 4>", n),
             "k_txframe64": kern("ofdm::k_txframe64<6, true>", n)},
    "cfg4": {"frames": n4, "slot_bytes_per_frame": info["cfg4_capture_bytes"] / n4,
             "k_sc_stream": kern("ofdm::k_sc_stream<2>", n4), "k_rxframe1024": kern("ofdm::k_rxframe1024<6, true>", n4, needed_bytes_per_frame=9 * 8192)},
    "cfg5": {"symbols": n5, "k_demod4096": kern("ofdm::k_demod4096<8, true, false>", n5), "k_tx4096": kern("ofdm::k_tx4096<true>", n5)},
}
c3 = out["cfg3"]
c3["chain_stated_placement_bytes_per_frame"] = round(c3["k_sc80_stated_placement"]["read_bytes_per_frame"] + c3["k_rxframe64"]["read_bytes_per_frame"]
                                                     + c3["k_rxframe64"]["write_bytes_per_frame"], 1)
out["per_kernel"] = {   # flat view for the bench blocks' `roofline.traffic` (tools/bench_cfg3._traffic)
    "k_sc80": {k: c3["k_sc80_stated_placement"][k] for k in ("read_bytes_per_frame", "write_bytes_per_frame")},
    "k_rxframe64": {k: c3["k_rxframe64"][k] for k in ("read_bytes_per_frame", "write_bytes_per_frame")},
    "k_sc_stream": {k: out["cfg4"]["k_sc_stream"][k] for k in ("read_bytes_per_frame", "write_bytes_per_frame")},
    "k_rxframe1024": {k: out["cfg4"]["k_rxframe1024"][k] for k in ("read_bytes_per_frame", "write_bytes_per_frame")},
}
print(json.dumps(out, indent=1))
