"""Practical HBM ceiling of the box: a 10 GiB torch copy (read + write) and reduction (read only), for comparison with the
kernels' physical traffic (profiles/r01_pmc_traffic.json).  Example (a box where bench.py gave 709 G samples/s, i.e.
k_demod64 at 4.9 TB/s physical): copy 4.66 TB/s, sum 3.99 TB/s."""
import torch, time, sys
sys.path.insert(0,'/root/repo')
x=torch.empty(10*1024**3//4, dtype=torch.float32, device='cuda'); y=torch.empty_like(x)
x.fill_(1.0); torch.cuda.synchronize()
for name,fn in (("copy", lambda: y.copy_(x)), ("sum", lambda: x.sum())):
    fn(); torch.cuda.synchronize()
    e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): fn()
    e1.record(); torch.cuda.synchronize()
    ms=e0.elapsed_time(e1)/10
    nbytes = x.numel()*4*(2 if name=="copy" else 1)
    print(name, round(ms,3), "ms", round(nbytes/ms/1e9,2), "TB/s")
