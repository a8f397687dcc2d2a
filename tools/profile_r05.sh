#!/bin/bash
# Round-5 rocprofv3 passes; run on the GPU box from the repo root:  bash tools/profile_r04.sh [bench|mv|c5 ...]
# Kernel trace + stats per workload, each PMC group in its own pass (never combined with other trace domains).
set -u
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_r05
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
export C5_ONLY_FULL=1
WORKLOADS=${@:-bench mv c5}
for w in $WORKLOADS; do
  case $w in
    bench) CMD="python3 $REPO/bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-extra --busy-seconds 0.2" ;;
    mv)    CMD="python3 $REPO/tools/bench_mv.py" ;;
    c5)    CMD="python3 $REPO/tools/bench_c5.py" ;;
  esac
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/${w}_trace" -- $CMD > "$OUT/${w}_trace.log" 2>&1
  echo "$w trace done"
  i=0
  for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS"; do
    i=$((i+1))
    rocprofv3 --pmc $grp --output-format csv -d "$OUT/${w}_pmc$i" -- $CMD > "$OUT/${w}_pmc$i.log" 2>&1
    echo "$w pmc pass $i done"
  done
  cd "$REPO"
  python3 tools/pmc_summary.py "$OUT/${w}_summary.json" "$OUT/${w}_trace" "$OUT/${w}_pmc1" "$OUT/${w}_pmc2" "$OUT/${w}_pmc3" "$OUT/${w}_pmc4" > "$OUT/${w}_summary.txt" 2>&1
  python3 tools/kstats.py "$OUT/${w}_trace" 16 > "$OUT/${w}_kernel_stats.txt" 2>&1
  # keep the merged output small: the raw per-dispatch CSVs stay on the box
  find "$OUT" -name "*counter_collection.csv" -delete; find "$OUT" -name "*kernel_trace.csv" -size +20M -delete
  cat "$OUT/${w}_summary.txt"
  cd /tmp
done
