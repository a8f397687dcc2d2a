#!/bin/bash
# The driver's bench command, then the one-rank rehearsal of the sharded path at the shard size of c3; prints the figures DESIGN.md quotes.
mkdir -p gpurun_out/r05
python bench.py --steps 20 --warmup 5 > gpurun_out/r05/bench_final.json 2> gpurun_out/r05/bench_final.err || exit 1
python bench.py --gpus 1 --rehearse-sharded --samples-total 125000 --steps 20 --warmup 5 > gpurun_out/r05/bench_shard_final.json 2> gpurun_out/r05/bench_shard_final.err || exit 1
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r05/bench_final.json").read().strip().splitlines()[-1])
print("c2:", d["value"], d["ms_per_step"], d["roofline"])
x = d["extra"]
print("c5 update ms:", x["c5_mmcorrnmf"]["update_ms_median_after_warmup"], "| c4 us/step:", x["c4_mvnmf"]["us_per_step"], "| weighted:", x["c2_weighted_step"]["us_per_step"])
print("time_to_kl:", {k: d["time_to_kl"].get(k) for k in ("gpu_loop_seconds", "gpu_fit_seconds_runs", "cpu_seconds")}, "| parity:", d["parity"])
s = json.loads(open("gpurun_out/r05/bench_shard_final.json").read().strip().splitlines()[-1])
print("shard:", s["value"], s["ms_per_step"], s["roofline"]["frac"], s["one_gpu_same_problem_value"], s["speedup_vs_one_gpu_same_problem"])
print(s["config"]["exchange"]["timeline_us_per_rank"])
PY
