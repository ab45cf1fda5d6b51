#!/bin/bash
# Same-box A/B (working tree vs _ab_base = a build of HEAD) of the R x 64 kernels: tools/lab/ab_mid.sh [N ...]
Ns=${@:-1024 2048}
for rep in 1 2; do
for d in . _ab_base; do
  echo "== $d rep $rep"
  (cd $d && timeout -k 10 200 python tools/bench_shapes.py 27 5 6 $Ns 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        r=json.loads(l); print(r['n_fft'], 'tx %.4f rx %.4f enc %.4f'%(r['tx_ms'],r.get('rx_ms',0),r['encode_ms']))
")
done; done
