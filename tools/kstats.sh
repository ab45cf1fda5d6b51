#!/bin/bash
# rocprofv3 kernel-trace summary of one python tool run: tools/kstats.sh <tag> <script.py> [args...]  -> gpurun_out/<tag>_kernel_stats.csv
# (the profiler gets the program itself after "--": no shell / env wrapper between rocprofv3 and python)
tag=$1; shift
out="$PWD/gpurun_out"; mkdir -p "$out"
export TMPDIR=/tmp
rm -rf /tmp/kst_$tag
( cd /tmp && timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kst_$tag -- python3 "$OLDPWD/$1" "${@:2}" > "$out/${tag}_run.log" 2> "$out/${tag}_rocprof.err" ) || exit 2
f=$(find /tmp/kst_$tag -name "*kernel_stats.csv" | head -1)
test -n "$f" || exit 3
( head -1 "$f"; grep "ofdm::" "$f" ) > "$out/${tag}_kernel_stats.csv"
cut -d, -f1-4 "$out/${tag}_kernel_stats.csv" | cut -c1-150
