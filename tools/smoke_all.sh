#!/bin/bash
# Runs every experiment / probe script under tools/ once, each under its own timeout, and records which still run against
# the current ABI (VERDICT r4 housekeeping): `bash tools/smoke_all.sh <out dir under gpurun_out>` on the GPU box. A script
# that times out after printing results is "ran (cut at the limit)"; one that raises is listed with its last line.
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$1
mkdir -p "$OUT"
: > "$OUT/status.txt"
for f in "$R"/tools/exp_*.py "$R"/tools/probe_*.py "$R"/tools/bench_gemm.py; do
  n=$(basename "$f" .py)
  timeout -k 5 ${SMOKE_LIMIT:-50} python "$f" > "$OUT/$n.log" 2>&1
  rc=$?
  lines=$(grep -c . "$OUT/$n.log")
  if [ $rc -eq 0 ]; then s="ok"; elif [ $rc -eq 124 ] || [ $rc -eq 137 ]; then s="cut at the limit after $lines lines"; else s="FAILED rc=$rc: $(tail -1 "$OUT/$n.log" | cut -c1-160)"; fi
  echo "$n: $s" | tee -a "$OUT/status.txt"
done
