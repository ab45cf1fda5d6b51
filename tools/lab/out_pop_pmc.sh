#!/bin/bash
# Counter passes over tools/lab/out_pop_probe.py (one group per rocprofv3 run, nothing but --pmc): what differs between an output buffer of
# the fast population and one of the slow population of k_demod64?  Results: gpurun_out/prof/${ROUND}_outpop_<tag>.{json,tsv}
set -o pipefail
ROUND=${ROUND:-r05}
OUT="$PWD/gpurun_out/prof"; mkdir -p "$OUT"
export TMPDIR=/tmp
run() {
  local tag=$1; shift
  rm -rf "/tmp/outpop_$tag"
  ( cd /tmp && timeout -k 10 300 rocprofv3 --pmc "$@" --output-format csv -d "/tmp/outpop_$tag" -- python3 "$OLDPWD/tools/lab/out_pop_probe.py" > "$OUT/${ROUND}_outpop_$tag.log" 2>&1 ) || { echo "pass $tag failed"; tail -3 "$OUT/${ROUND}_outpop_$tag.log"; return; }
  grep "^{" "$OUT/${ROUND}_outpop_$tag.log" > "$OUT/${ROUND}_outpop_$tag.json"
  python3 tools/pmc_summary.py $(find "/tmp/outpop_$tag" -name "*counter_collection.csv") | grep "k_demod64" > "$OUT/${ROUND}_outpop_$tag.tsv"
  echo "outpop $tag done"
}
timeout -k 10 200 python3 tools/lab/out_pop_probe.py > "$OUT/${ROUND}_outpop_plain.json" 2> /dev/null; echo "plain done"
run utcl1 TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum
run wrstall TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum TCC_EA0_WRREQ_LEVEL_sum
run wrlat TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum
run wrmix TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_DRAM_sum TCC_EA0_RDREQ_DRAM_sum
run rdstall TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_RDREQ_LEVEL_sum TCC_TAG_STALL_sum TCC_IB_STALL_sum
run utcl2 GRBM_UTCL2_BUSY GRBM_GUI_ACTIVE
echo "all done"
