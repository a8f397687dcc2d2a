"""Print the top rows of a rocprofv3 --stats kernel_stats.csv found under the given directory."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[: int(sys.argv[2]) if len(sys.argv) > 2 else 14]:
    print(r["Name"][:72].ljust(72), r["Calls"].rjust(6), "avg_us=%8.1f" % (float(r["AverageNs"]) / 1e3), "pct=%s" % r["Percentage"])
