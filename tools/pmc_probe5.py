"""Workload for the round-5 rocprofv3 --pmc passes (tools/pmc_round5.sh).  First three CALIBRATION launches whose byte counts are known
exactly -- the read-only probe k_read_probe in k_demod64's access pattern (8 B per lane, 512 B of every 640-byte symbol: the pattern of
every frame kernel), over whole symbols, and with unit-stride 16-byte loads -- then every kernel the bench times, on packet frames at
sizes beyond the 256 MiB Infinity Cache.  Prints the byte counts the summaries are divided by."""
import json, math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ofdm_amd import api
from tools import bench_cfg3, bench_large_n

n = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
info = {"frames_cfg2_cfg3": n}
ctx = api.Context(n_fft=64, modulation=api.QAM64, guard_bands=True)
g = torch.Generator(device="cuda"); g.manual_seed(1)
pay = torch.randint(0, 256, (n, 16 * 36), dtype=torch.uint8, device="cuda", generator=g)
x2 = ctx.prefix_block(ctx.encode_block(ctx.modulate(pay.view(-1)).view(-1, 48))).view(n, 16 * 80)  # cfg2 shape: 2.7 GB
out2 = torch.empty((n, 16 * 36), dtype=torch.uint8, device="cuda")
v = torch.view_as_complex(torch.randn((n * 16, 64, 2), device="cuda")).contiguous()         # FFT calibration (reads and writes n*16*512 B)
vo = torch.empty_like(v)
x3, p3 = bench_cfg3.synth(api, torch, ctx, n)
for _ in range(2):
    for pattern in (0, 1, 2):
        ctx.hbm_read_probe(x2, pattern)      # k_read_probe<pattern>: 512 / 640 / 640 B per symbol, nothing written
    ctx.fft(v, out=vo)
    ctx.rx_demod(x2, syms_per_frame=16, out=out2)
    ctx.decode_batch(x3, max_symbols=16)     # product chain: k_sc80 + k_sc_post + k_rx_prepare + k_rxframe64<finish>
    ctx.set_tuning("no_sc80", 1); ctx.set_tuning("sc_first_lags", 0)
    ctx.sc_correlate(x3)                     # the f32 filter kernel that computes every lag: k_sc_cf<256,2,4>
    ctx.set_tuning("no_sc80", 0); ctx.set_tuning("sc_first_lags", 576)
    ctx.encode_batch(p3)
torch.cuda.synchronize()
info.update(probe_symbols=n * 16, fft_bytes_each_way=n * 16 * 512, demod_alg_read=n * 16 * 640, demod_write=n * 16 * 36,
            cfg3_capture_bytes=n * 2176 * 8, cfg3_payload_bytes=n * 560, tx_frame_bytes=n * 2080 * 8)
del v, vo, x2, out2, x3, p3
torch.cuda.empty_cache()
# late packets + empty slots: k_sc80 where it cannot stop early
xl, pl = bench_cfg3.synth(api, torch, ctx, n, span=bench_cfg3.LATE_SPAN, seed=31, max_delay=bench_cfg3.LATE_SPAN - 2080 - 63,
                          noise_only=bench_cfg3.LATE_NOISE_ONLY)
xn = torch.view_as_complex(torch.randn((n, bench_cfg3.SPAN, 2), dtype=torch.float32, device="cuda") * 0.004)
for _ in range(2):
    ctx.sc_correlate(xl)
    ctx.sc_correlate(xn)
torch.cuda.synchronize()
info.update(late_capture_bytes=n * bench_cfg3.LATE_SPAN * 8, noise_capture_bytes=n * bench_cfg3.SPAN * 8)
del xl, pl, xn
torch.cuda.empty_cache()
# cfg4: a 16 Ki-frame ring (2.4 GB)
c4 = api.Context(n_fft=1024, modulation=api.QAM64, guard_bands=True, ecc=api.ECC_HAMMING74)
n4 = 16384
pay4 = torch.randint(0, 256, (n4, bench_large_n.CFG4_NBYTES), dtype=torch.uint8, device="cuda", generator=g)
tx4 = c4.encode_batch(pay4)
d = torch.randint(1, 65, (n4,), device="cuda", generator=g, dtype=torch.int32)
fd = (torch.rand((n4,), device="cuda", generator=g, dtype=torch.float64) * 1.9 - 0.95) * math.pi / c4.S
x4 = c4.channel_batch(tx4, snr_db=40.0, seed=44, delay=d, f_delta=fd, span=tx4.shape[1] + 256)
for _ in range(2):
    c4.decode_batch(x4, max_symbols=4)
torch.cuda.synchronize()
info.update(frames_cfg4=n4, cfg4_capture_bytes=n4 * x4.shape[1] * 8)
del tx4, x4
torch.cuda.empty_cache()
# cfg5: 32 Ki symbols (1.3 GB)
c5 = api.Context(n_fft=4096, modulation=api.QAM256, guard_bands=True)
n5 = 32768
pay5 = torch.randint(0, 256, (n5 * c5.bytes_per_symbol,), dtype=torch.uint8, device="cuda", generator=g)
out5 = torch.empty((1, pay5.numel()), dtype=torch.uint8, device="cuda")
for _ in range(2):
    x5 = c5.tx_symbols(pay5)
    c5.rx_demod(x5.view(1, -1), syms_per_frame=n5, out=out5)
torch.cuda.synchronize()
info.update(symbols_cfg5=n5, cfg5_sample_bytes=n5 * 5120 * 8, cfg5_payload_bytes=int(pay5.numel()))
print(json.dumps(info))
