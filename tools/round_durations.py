"""Durations of the lockstep evaluation rounds in launch order, from a rocprofv3 --kernel-trace csv directory:
python tools/round_durations.py <trace dir> [kernel name part]"""
import csv, glob, os, sys
rows = []
for f in glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True):
    with open(f) as fh:
        rows += list(csv.DictReader(fh))
part = sys.argv[2] if len(sys.argv) > 2 else "ls_eval"
rows = [r for r in rows if part in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
print(len(d), "launches; durations in us, in order:")
print(" ".join(f"{x:.0f}" for x in d))
