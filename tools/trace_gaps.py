"""From a rocprofv3 --kernel-trace CSV: for the steady-state stretch of fused / tail dispatches, the durations and the gaps
(end of one dispatch -> start of the next) in us.  usage: trace_gaps.py <trace dir>"""
import csv, glob, sys
import numpy as np
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
ev = [(r["Kernel_Name"], int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows]
def kind(n):
    return "fused" if "fused_kernel" in n else "tailp2p" if "tail_p2p" in n else "tail" if "tail_kernel" in n else "other"
out = {}
for (n0, s0, e0), (n1, s1, e1) in zip(ev, ev[1:]):
    k0, k1 = kind(n0), kind(n1)
    if "other" in (k0, k1):
        continue
    out.setdefault(f"gap {k0}->{k1}", []).append((s1 - e0) / 1e3)
    out.setdefault(f"dur {k0}", []).append((e0 - s0) / 1e3)
    out.setdefault(f"period {k0}->{k1} (start to start)", []).append((s1 - s0) / 1e3)
for k in sorted(out):
    v = np.array(out[k])
    print(f"{k:44s} n={len(v):6d} median {np.median(v):8.2f} mean {v.mean():8.2f} p10 {np.quantile(v, 0.1):8.2f} p90 {np.quantile(v, 0.9):8.2f}")
