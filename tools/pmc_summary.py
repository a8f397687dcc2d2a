"""Average per dispatch of every counter of every kernel in the rocprofv3 --pmc CSVs under the given directories
(one directory per pass), plus the kernel-trace durations when a trace directory is given first:

    python tools/pmc_summary.py OUT.json TRACE_DIR PMC_DIR [PMC_DIR ...]
"""
import csv, glob, json, sys
from collections import defaultdict

out_path, trace_dir, pmc_dirs = sys.argv[1], sys.argv[2], sys.argv[3:]
summary = {"kernels": {}}


def short(name):
    return name.split("(")[0].replace("void ", "")


for f in glob.glob(trace_dir + "/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = summary["kernels"].setdefault(short(r["Name"]), {})
        k["calls"] = int(r["Calls"])
        k["avg_us"] = float(r["AverageNs"]) / 1e3
        k["pct_of_gpu_time"] = float(r["Percentage"])
for d in pmc_dirs:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        acc = defaultdict(lambda: defaultdict(list))
        for r in csv.DictReader(open(f)):
            acc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, cs in acc.items():
            for c, v in cs.items():
                summary["kernels"].setdefault(k, {}).setdefault("counters", {})[c] = {"n": len(v), "avg": sum(v) / len(v)}
# HBM bytes per launch as MI355X_MICROARCH.md prescribes for gfx950: FETCH_SIZE and WRITE_SIZE count 32-byte... (kept raw here;
# the callers apply FETCH_SIZE * 1024 * 2 (gfx950 correction) + WRITE_SIZE * 1024, as profiles/pmc_summary.json documents)
for k, rec in summary["kernels"].items():
    c = rec.get("counters", {})
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        rec["hbm_bytes_per_launch"] = c["FETCH_SIZE"]["avg"] * 1024 * 2 + c["WRITE_SIZE"]["avg"] * 1024
json.dump(summary, open(out_path, "w"), indent=1)
top = sorted(summary["kernels"].items(), key=lambda kv: -kv[1].get("pct_of_gpu_time", 0))[:12]
for k, rec in top:
    c = rec.get("counters", {})
    line = f"{k[:70]:70s} calls={rec.get('calls', 0):6d} avg_us={rec.get('avg_us', 0):9.1f} pct={rec.get('pct_of_gpu_time', 0):5.1f}"
    for name in ("SQ_INSTS_VALU", "SQ_INSTS_MFMA", "SQ_INSTS_LDS", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE"):
        if name in c:
            line += f" {name}={c[name]['avg']:.4g}"
    if "hbm_bytes_per_launch" in rec:
        line += f" HBM_MB={rec['hbm_bytes_per_launch'] / 1e6:.1f}"
    print(line)
