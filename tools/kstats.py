"""Per-kernel duration statistics from a rocprofv3 kernel trace csv directory."""
import csv, glob, sys, collections, os
for f in glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"][:70]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
        v.sort()
        print("%-72s n %5d  mean %8.1f us  median %8.1f  p90 %8.1f  max %8.1f" % (k, len(v), sum(v) / len(v), v[len(v) // 2], v[int(len(v) * 0.9)], v[-1]))
