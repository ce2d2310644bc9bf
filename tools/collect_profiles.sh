#!/bin/bash
# Collects the rocprofv3 evidence of a round ON THE GPU BOX (run through gpurun from the repo root):
#   bash tools/collect_profiles.sh <out dir under gpurun_out> <what>
# what = stats_unet | stats_sr4 | stats_swinir | pmc_unet | pmc_swinir
# The summaries under profiles/ are then written in the build container by tools/kernel_stats_summary.py and
# tools/pmc_summary.py. One rocprofv3 pass per counter group (FETCH_SIZE and WRITE_SIZE do not fit one pass).
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$1
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
COMMON="--no-cpu-baseline --no-secondary --no-profile-gemms"
case "$2" in
  stats_unet)   ARGS="$COMMON --steps 20 --warmup 5" ;;
  stats_sr4)    ARGS="$COMMON --task sr --sr-factor 4 --steps 5 --warmup 2" ;;
  stats_swinir) ARGS="$COMMON --arch swinir --task sr --sr-factor 2 --steps 5 --warmup 2" ;;
  pmc_unet)     ARGS="$COMMON --pmc-twin --steps 2 --warmup 1" ;;
  pmc_swinir)   ARGS="$COMMON --arch swinir --task sr --sr-factor 2 --pmc-twin --steps 2 --warmup 1" ;;
  *) echo "unknown: $2"; exit 2 ;;
esac
case "$2" in
  stats_*)
    rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/$2" -o q -- python3 "$R/bench.py" $ARGS > "$OUT/$2.log" 2>&1 || exit 1
    ;;
  pmc_*)
    for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
      name=${grp%% *}
      rocprofv3 --pmc $grp --output-format csv -d "$OUT/$2/pmc_$name" -o pmc -- python3 "$R/bench.py" $ARGS > "$OUT/$2_$name.log" 2>&1 || exit 1
      echo "pass $name done"
    done
    ;;
esac
grep -h '^{' "$OUT"/$2*.log | head -3 | cut -c1-300
