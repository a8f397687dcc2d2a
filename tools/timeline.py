"""Development aid: print the kernel timeline (start offset, duration, stream/queue) of a window of a rocprofv3 kernel trace."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
lo = int(sys.argv[2]) if len(sys.argv) > 2 else len(rows) // 2
n = int(sys.argv[3]) if len(sys.argv) > 3 else 40
t0 = int(rows[lo]["Start_Timestamp"])
for r in rows[lo:lo + n]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"{(s - t0) / 1e3:9.1f} us  +{(e - s) / 1e3:7.1f} us  q={r.get('Queue_Id', '?'):>3}  {r['Kernel_Name'][:70]}")
