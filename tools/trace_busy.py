"""From a rocprofv3 --kernel-trace CSV: how much of the wall time between the first and the last dispatch the GPU had at least one
kernel running (union of the dispatch intervals), the largest idle gaps and what ran around them.  usage: trace_busy.py <dir> [skip_fraction]"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
rows = rows[int(len(rows) * skip):]  # steady state: the second half of the run by default
busy, cur_s, cur_e, gaps = 0, None, None, []
prev = None
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if cur_e is None:
        cur_s, cur_e = s, e
    elif s <= cur_e:
        cur_e = max(cur_e, e)
    else:
        busy += cur_e - cur_s
        gaps.append((s - cur_e, prev, r["Kernel_Name"][:60]))
        cur_s, cur_e = s, e
    prev = r["Kernel_Name"][:60]
busy += cur_e - cur_s
wall = cur_e - int(rows[0]["Start_Timestamp"])
print(f"{len(rows)} dispatches over {wall / 1e6:.2f} ms: GPU busy {busy / 1e6:.2f} ms = {busy / wall:.3f}; {len(gaps)} idle gaps, {sum(g[0] for g in gaps) / 1e6:.2f} ms in all")
by = {}
for g, a, b in gaps:
    by.setdefault((a, b), []).append(g)
for (a, b), v in sorted(by.items(), key=lambda kv: -sum(kv[1]))[:12]:
    print(f"  {sum(v) / 1e3:9.1f} us in {len(v):4d} gaps (mean {sum(v) / len(v) / 1e3:6.1f} us)  after {a}  before {b}")
