#!/bin/bash
# Same-box A/B (working tree vs _ab_base = a build of HEAD) of the N = 64 Schmidl-Cox kernels: per delay band the time of the one-launch and
# two-launch searches (tools/late_diag.py)
for rep in 1 2; do
for d in . _ab_base; do
  echo "== $d rep $rep"
  (cd $d && timeout -k 10 200 python tools/late_diag.py 262144 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        r=json.loads(l); print(r['band'], 'one launch %.4f  two launches %.4f  slow %d'%(r['first0_ms'], r['first384_ms'], r['first0_slow']))
")
done; done
