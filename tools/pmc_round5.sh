#!/bin/bash
# Round-5 PMC passes (run on the GPU box through gpurun, from the repo root): raw memory-side counters of the L2 instead of the derived
# FETCH_SIZE (which rounds 3 / 4 found unreliable for 8-byte-per-lane loads at arbitrary offsets), each group in its OWN rocprofv3 run
# with nothing but --pmc (MI355X_MICROARCH.md, rocprofv3 PMC slots), the read group TWICE (reproducibility), all on tools/pmc_probe5.py,
# whose first launches are calibration kernels with exactly known byte counts.
set -o pipefail
ROUND=${ROUND:-r05}
OUT="$PWD/gpurun_out/prof"; mkdir -p "$OUT"
export TMPDIR=/tmp
run() { # tag, counters...
  local tag=$1; shift
  rm -rf "/tmp/prof5_$tag"
  ( cd /tmp && timeout -k 10 500 rocprofv3 --pmc "$@" --output-format csv -d "/tmp/prof5_$tag" -- python3 "$OLDPWD/tools/pmc_probe5.py" > "$OUT/${ROUND}_pmc5_$tag.log" 2>&1 ) || { echo "pass $tag failed"; tail -5 "$OUT/${ROUND}_pmc5_$tag.log"; exit 4; }
  python3 tools/pmc_summary.py $(find "/tmp/prof5_$tag" -name "*counter_collection.csv") > "$OUT/${ROUND}_pmc5_$tag.tsv"
  echo "pmc $tag done"
}
run rd_a TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_BUBBLE_sum
run rd_b TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_BUBBLE_sum
run l2 TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum
run wr TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_STALL_sum
run fetch FETCH_SIZE
run write WRITE_SIZE
echo "all done"
