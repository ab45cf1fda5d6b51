"""Workload for the rocprofv3 --pmc passes (tools/profile_round.sh): one calibration kernel with a known byte count in the
hot kernels' 8-byte-per-lane access pattern (batched FFT: reads and writes n*512 B), then every kernel the bench times, on
PACKET frames (the library's TX through its GPU channel model), at sizes beyond the 256 MiB Infinity Cache:
  cfg2  k_demod64                      cfg3  k_sc_cf + k_sc_post + k_rx_prepare + k_rxframe64, and the one-pass k_sc_cf<..,6,true>
  cfg4  k_sc_stream (L = 1280, every lag) + k_rx_prepare + k_rxframe1024<finish>        cfg5  k_tx4096, k_demod4096
  mid   k_tx_mid, k_demod_mid, k_txframe_mid at N = 512 and 2048
Prints the byte counts the summaries are divided by."""
import json, math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ofdm_amd import api
from tools import bench_cfg3, bench_large_n

n = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
info = {"frames_cfg2_cfg3": n}
ctx = api.Context(n_fft=64, modulation=api.QAM64, guard_bands=True)
v = torch.view_as_complex(torch.randn((n * 16, 64, 2), device="cuda")).contiguous()         # FFT calibration
vo = torch.empty_like(v)
g = torch.Generator(device="cuda"); g.manual_seed(1)
pay = torch.randint(0, 256, (n, 16 * 36), dtype=torch.uint8, device="cuda", generator=g)
x2 = ctx.prefix_block(ctx.encode_block(ctx.modulate(pay.view(-1)).view(-1, 48))).view(n, 16 * 80)  # cfg2 shape
out2 = torch.empty((n, 16 * 36), dtype=torch.uint8, device="cuda")
x3, p3 = bench_cfg3.synth(api, torch, ctx, n)
for _ in range(2):
    ctx.fft(v, out=vo)
    ctx.rx_demod(x2, syms_per_frame=16, out=out2)
    ctx.set_tuning("sc_first_lags", 0)
    ctx.sc_correlate(x3)                     # the kernel that computes every lag: k_sc_cf<256,2,4,0> over every frame
    ctx.set_tuning("sc_first_lags", 576)
    ctx.decode_batch(x3, max_symbols=16)     # product path: k_sc_cf<128,1,5,0> over the first lags + k_sc_cf<256,2,4,0> over the (empty) list
    ctx.set_tuning("one_pass_rx", 1)
    ctx.decode_batch(x3, max_symbols=16)
    ctx.set_tuning("one_pass_rx", 0)
    ctx.encode_batch(p3)
torch.cuda.synchronize()
info.update(fft_bytes_each_way=n * 16 * 512, demod_alg_read=n * 16 * 640, demod_write=n * 16 * 36, cfg3_capture_bytes=n * 2176 * 8,
            cfg3_payload_bytes=n * 560, tx_frame_bytes=n * 2080 * 8)
del v, vo, x2, out2, x3, p3
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
    c4.decode_batch(x4, max_symbols=4)   # every lag (the headline of the config-4 block); the bounded search runs the same kernels on fewer tiles
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
del x5, pay5, out5
torch.cuda.empty_cache()
# the R x 64 family (kernels_mid.hip): N = 512 (wave-local) and N = 2048 (barriers), 2^27 samples each: k_tx_mid, k_demod_mid,
# and k_txframe_mid on 16-symbol frames
for nn in (512, 2048):
    cm = api.Context(n_fft=nn, modulation=api.QAM64, guard_bands=True)
    ns = (1 << 27) // cm.S // 8 * 8
    paym = torch.randint(0, 256, (ns * cm.bytes_per_symbol,), dtype=torch.uint8, device="cuda", generator=g)
    outm = torch.empty((ns // 8, 8 * cm.bytes_per_symbol), dtype=torch.uint8, device="cuda")
    nb = 16 * cm.bytes_per_symbol - 16
    nfr = (1 << 27) // cm.frame_samples(nb)
    payf = torch.randint(0, 256, (nfr, nb), dtype=torch.uint8, device="cuda", generator=g)
    for _ in range(2):
        xm = cm.tx_symbols(paym, ns)
        cm.rx_demod(xm.view(ns // 8, 8 * cm.S), 8, out=outm)
        fo = cm.encode_batch(payf)
    torch.cuda.synchronize()
    info[f"mid_{nn}"] = {"symbols": ns, "sample_bytes": ns * cm.S * 8, "payload_bytes": int(paym.numel()), "frames": nfr,
                         "frame_bytes": nfr * cm.frame_samples(nb) * 8}
    del xm, paym, outm, fo, payf
    torch.cuda.empty_cache()
print(json.dumps(info))
