"""Start / end of every trace launch and blend of the timed region of a short burst (rocprofv3 --kernel-trace of bench.py,
profiles/r04/evidence/timelines/*.csv): how 16 overlapped 64-workgroup launches tile the 512 workgroup slots.
usage: python profiles/r04/scripts/analyse_timelines.py profiles/r04/evidence/timelines/s20_kernel_trace.csv 20"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2])
ks = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"])) for r in rows)
tr = [k for k in ks if "TraceQueue" in k[2]][-n:]
t0 = tr[0][0]
rs = [k for k in ks if "Resolve" in k[2] and k[0] >= t0]
print("launch  start_us   end_us   dur_us  workgroups")
for i, (s, e, _, g) in enumerate(tr):
    print("%5d %9.1f %9.1f %8.1f %6d" % (i + 1, (s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, g))
ends = sorted((e - t0) / 1e3 for _, e, _, _ in tr)
print("trace launches complete at (us):", " ".join("%.0f" % x for x in ends))
print("blend kernels start(duration) us:", " ".join("%.0f(%.0f)" % ((s - t0) / 1e3, (e - s) / 1e3) for s, e, _, _ in rs))
print("last trace launch ends %.0f us, last blend ends %.0f us" % (ends[-1], (rs[-1][1] - t0) / 1e3))
