#!/bin/bash
# Same-box A/B (working tree vs _ab_base = a build of HEAD) of the config-4 chain (tools/prof_clean.py cfg4_chain / cfg4_late, 16 384 frames)
for rep in 1 2; do
for d in . _ab_base; do
  echo "== $d rep $rep"
  (cd $d && for w in cfg4_chain cfg4_late; do timeout -k 10 200 python tools/prof_clean.py $w 16384 2>/dev/null | cut -c1-90; done)
done; done
