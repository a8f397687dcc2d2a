"""Where a step of a small cohort goes on the per-step path: run under rocprofv3 --kernel-trace (csv), then summarise the
trace with `python tools/timeline_small.py --summarise <trace dir>`: per kernel the mean duration, and the mean gap from
the end of a kernel to the start of the next one on the stream.

    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl -- python3 tools/timeline_small.py 192 5
"""
import csv, glob, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

if sys.argv[1] == "--summarise":
    rows = []
    for f in glob.glob(os.path.join(sys.argv[2], "**", "*kernel_trace.csv"), recursive=True):
        with open(f) as fh:
            rows += list(csv.DictReader(fh))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    rows = rows[len(rows) // 2:]  # (the second half: steady state)
    dur, gap, prev = {}, {}, None
    for r in rows:
        name = r["Kernel_Name"].split("(")[0][-48:]
        s, t = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        dur.setdefault(name, []).append(t - s)
        if prev is not None:
            gap.setdefault(name, []).append(s - prev)
        prev = t
    span = (int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])) / 1e3
    print(f"{len(rows)} dispatches over {span:.1f} us")
    for name in dur:
        g = gap.get(name, [0])
        print(f"{name:50s} n={len(dur[name]):5d} duration {sum(dur[name]) / len(dur[name]) / 1e3:6.2f} us, gap before it {sum(g) / len(g) / 1e3:6.2f} us")
    sys.exit(0)

from salamander_amd import Engine, synthetic
N, K = int(sys.argv[1]), int(sys.argv[2])
tiles = int(sys.argv[3]) if len(sys.argv) > 3 else 0
X, W0, H0 = synthetic.synthetic_problem(96, N, K, seed=0)
e = Engine(N, 96, K)
e.set_small_cohort_tiles(tiles)
e.upload_X(X), e.upload_W(W0), e.upload_H(H0)
e.kl_step(400, 0)
e.sync()
e.close()
