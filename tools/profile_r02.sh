#!/bin/bash
# Round-2 rocprofv3 passes; run on the GPU box from the repo root:  bash tools/profile_r02.sh
# Kernel trace + stats per workload, each PMC group in its own pass (never combined with other trace domains).
set -u
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_r02
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $REPO/bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-extra --busy-seconds 0.2"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/bench_trace" -- $BENCH > "$OUT/bench_trace.log" 2>&1
echo "bench trace done"
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  name=$(echo $grp | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $grp --output-format csv -d "$OUT/bench_pmc_$name" -- $BENCH > "$OUT/bench_pmc_$name.log" 2>&1
  echo "bench pmc $name done"
done
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/mv_trace" -- python3 $REPO/tools/bench_mv.py > "$OUT/mv_trace.log" 2>&1
echo "mv trace done"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/c5_trace" -- python3 $REPO/tools/bench_c5.py > "$OUT/c5_trace.log" 2>&1
echo "c5 trace done"
for grp in "FETCH_SIZE" "WRITE_SIZE"; do
  rocprofv3 --pmc $grp --output-format csv -d "$OUT/mv_pmc_$grp" -- python3 $REPO/tools/bench_mv.py > "$OUT/mv_pmc_$grp.log" 2>&1
  echo "mv pmc $grp done"
done
cd "$REPO"
for w in bench mv c5; do
  echo "== $w kernel stats"; python3 tools/kstats.py "$OUT/${w}_trace" 16
done > "$OUT/kernel_stats.txt" 2>&1
python3 tools/timeline.py "$OUT/mv_trace" 400 40 > "$OUT/mv_timeline.txt" 2>&1
cat "$OUT/kernel_stats.txt"; cat "$OUT/mv_timeline.txt"
