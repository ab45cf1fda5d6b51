"""Small workload for rocprofv3 --pmc passes (FETCH_SIZE / WRITE_SIZE): one calibration kernel with a known byte
count in the same access pattern (batched FFT: reads and writes n*512 B with 8-B-per-lane accesses), the config-2
RX-demod kernel, and the Schmidl-Cox kernel on config-3 shaped frames.  Inputs exceed the 256 MiB Infinity Cache."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ofdm_amd import api

n = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
ctx = api.Context(n_fft=64, modulation=api.QAM64, guard_bands=True)
x = torch.view_as_complex(torch.randn((n, 16 * 80, 2), device="cuda") * 0.1).contiguous()   # cfg2 shape
out = torch.empty((n, 16 * 36), dtype=torch.uint8, device="cuda")
v = torch.view_as_complex(torch.randn((n * 16, 64, 2), device="cuda")).contiguous()         # FFT calibration
vo = torch.empty_like(v)
y = torch.view_as_complex(torch.randn((n, 2176, 2), device="cuda") * 0.1).contiguous()      # cfg3 shape
for _ in range(2):
    ctx.fft(v, out=vo)
    ctx.rx_demod(x, syms_per_frame=16, out=out)
    ctx.sc_correlate(y)
torch.cuda.synchronize()
print("frames", n, "fft_bytes_each_way", n * 16 * 512, "demod_alg_read", n * 16 * 640, "demod_fetched_expected", n * 16 * 512,
      "demod_write", n * 16 * 36, "sc_read", n * 2176 * 8)
