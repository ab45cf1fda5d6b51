#!/bin/bash
# Same-box A/B of two builds of the library: the working tree against a copy of HEAD built under _ab_base/ (git archive HEAD
# ofdm_amd tools oracle include | tar -x -C _ab_base; python -c "import ofdm_amd.build as b; b.build()" there).
for rep in 1 2; do
for d in . _ab_base; do
  echo "== $d rep $rep"
  (cd $d && timeout -k 10 200 python tools/bench_shapes.py 27 5 6 64 512 1024 4096 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        r=json.loads(l); print(r['n_fft'], 'tx %.3f rx %.3f enc %.3f'%(r['tx_ms'],r['rx_ms'],r['encode_ms']))
")
  (cd $d && for w in cfg3_chain cfg3_late; do timeout -k 10 200 python tools/prof_clean.py $w 262144 2>/dev/null | cut -c1-120; done)
done; done
