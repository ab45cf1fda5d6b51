"""Per-frame section times (s_memtime ticks, wave 0) of the one-pass receive kernel, OFDM_SC_DEBUG=20..26:
DMA wait | phase 1 | coarse | fine + timing tail | channel estimate + data groups | finish | whole iteration."""
import os, sys, subprocess, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import torch
    from ofdm_amd import api
    from tools import bench_cfg3
    ctx = api.Context(n_fft=64, modulation=api.QAM64, guard_bands=True)
    x, payload = bench_cfg3.synth(api, torch, ctx, 65536, 2176)
    r = ctx.decode_batch(x, max_symbols=ctx.data_symbols(560))
    torch.cuda.synchronize()
    print(float(r["metric"].double().mean()))
else:
    names = ["dma_wait", "phase1", "coarse", "fine+timing", "chest+data", "finish", "iteration"]
    out = {}
    for i, nm in enumerate(names):
        env = dict(os.environ, OFDM_SC_DEBUG=str(20 + i), OFDM_ONE_PASS_RX="1")
        out[nm] = float(subprocess.run([sys.executable, __file__, "child"], env=env, capture_output=True, text=True).stdout.strip().splitlines()[-1])
    print(json.dumps(out))
