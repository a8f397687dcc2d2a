"""Condense the rocprofv3 CSVs written by tools/profile.sh into a short text + JSON summary."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

out = sys.argv[1]
summary = {}


def find(pattern):
    return sorted(glob.glob(os.path.join(out, pattern), recursive=True))


# ---- kernel stats
for f in find("trace/**/*kernel_stats.csv"):
    print("== kernel stats:", os.path.relpath(f, out))
    rows = list(csv.DictReader(open(f)))
    for r in rows[:12]:
        name = r.get("Name", "")[:90]
        print(f"  {name:90s} calls={r.get('Calls')} avg_ns={r.get('AverageNs')} total_ns={r.get('TotalDurationNs')} pct={r.get('Percentage')}")
        summary.setdefault("kernel_stats", []).append(
            {"name": r.get("Name"), "calls": int(r.get("Calls", 0)), "avg_ns": float(r.get("AverageNs", 0)), "pct": float(r.get("Percentage", 0))}
        )

# ---- counters: average per dispatch of each kernel
for f in find("pmc_*/**/*counter_collection.csv"):
    acc = defaultdict(lambda: defaultdict(list))
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    print("== counters:", os.path.relpath(f, out))
    for k, cs in acc.items():
        if not any(t in k for t in ("fused_kernel", "forward_kernel", "tail_kernel")):
            continue
        for c, v in cs.items():
            avg = sum(v) / len(v)
            print(f"  {k[:60]:60s} {c:32s} n={len(v):4d} avg={avg:.6g}")
            summary.setdefault("counters", {}).setdefault(k.split("(")[0], {})[c] = avg

json.dump(summary, open(os.path.join(out, "summary.json"), "w"), indent=1)
