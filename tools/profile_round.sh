#!/bin/bash
# Round profile set (run on the GPU box through gpurun, from the repo root; ROUND=r03 names the files):
#   1. bench.py as the driver runs it                      -> gpurun_out/prof/${ROUND}_bench.json
#   2. the same command under rocprofv3 --kernel-trace     -> ${ROUND}_bench_kernel_stats_ofdm_only.csv
#   3. separate --pmc SQ passes on tools/pmc_probe5.py     -> ${ROUND}_pmc_sq_summary.tsv  (never combined with a trace:
#      MI355X_MICROARCH.md, HBM + rocprofv3 sections).  The HBM-traffic passes (raw TCC_EA0 request counters) are
#      tools/pmc_round5.sh + tools/pmc_traffic5.py -> profiles/${ROUND}_pmc_traffic.json
# rocprofv3 writes its (large) traces under /tmp; only the summaries are copied back.
set -o pipefail
ROUND=${ROUND:-r05}
OUT="$PWD/gpurun_out/prof"; mkdir -p "$OUT"
export TMPDIR=/tmp
BENCH_ARGS=${BENCH_ARGS:-}
if [ -z "$SKIP_BENCH" ]; then
timeout -k 10 900 python3 bench.py --detail "$OUT/${ROUND}_bench_detail.json" $BENCH_ARGS > "$OUT/${ROUND}_bench.json" 2> "$OUT/${ROUND}_bench.err" || exit 1
echo "bench done"
# (--no-host: the host-buffer blocks launch k_demod64 / the decode chain on small chunks; without them k_demod64's average in this
#  trace is the average of the timed headline configuration only)
rm -rf /tmp/prof_kt
( cd /tmp && timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_kt -- python3 "$OLDPWD/bench.py" --no-host --detail /tmp/bench_detail_under_rocprof.json $BENCH_ARGS > "$OUT/${ROUND}_bench_under_rocprof.json" 2> "$OUT/${ROUND}_rocprof_kt.err" ) || exit 2
f=$(find /tmp/prof_kt -name "*kernel_stats.csv" | head -1)
test -n "$f" || exit 3
( head -1 "$f"; grep "ofdm::" "$f" ) > "$OUT/${ROUND}_bench_kernel_stats_ofdm_only.csv"
echo "kernel trace done"
fi
# 2b. clean per-configuration passes: ONE timed configuration per process (tools/prof_clean.py), so that a kernel's average is the
#     average of the measured launches only -- the every-lag search k_sc80, the config-3 / config-4 chains on the stated and on the late-packet placement, config 5
if [ -z "$SKIP_CLEAN" ]; then
for w in sc_every_lag cfg3_chain cfg3_late cfg4_chain cfg4_late cfg5; do
  rm -rf /tmp/prof_clean_$w
  ( cd /tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_clean_$w -- python3 "$OLDPWD/tools/prof_clean.py" $w > "$OUT/${ROUND}_clean_$w.json" 2> "$OUT/${ROUND}_clean_$w.err" ) || exit 6
  f=$(find /tmp/prof_clean_$w -name "*kernel_stats.csv" | head -1)
  test -n "$f" || exit 7
  ( head -1 "$f"; grep "ofdm::" "$f" ) > "$OUT/${ROUND}_clean_${w}_kernel_stats.csv"
  echo "clean $w done"
done
fi
if [ -n "$SKIP_PMC" ]; then echo "all done (no PMC passes)"; exit 0; fi
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS"; do
  tag=$(echo $grp | tr ' ' '_' | cut -c1-40)
  ( cd /tmp && timeout -k 10 400 rocprofv3 --pmc $grp --output-format csv -d "/tmp/prof_sq_$tag" -- python3 "$OLDPWD/tools/pmc_probe5.py" > "$OUT/${ROUND}_pmc_sq_$tag.log" 2>&1 ) || exit 5
  echo "pmc $tag done"
done
python3 tools/pmc_summary.py $(find /tmp/prof_sq_* -name "*counter_collection.csv") > "$OUT/${ROUND}_pmc_sq_summary.tsv"
echo "all done"
