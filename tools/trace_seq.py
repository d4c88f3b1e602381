"""Print the kernels of the LAST third of a rocprofv3 kernel trace in launch order (tuning aid)."""
import csv, glob, sys, re
d = sys.argv[1]
rows = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[2 * len(rows) // 3:]
t0 = int(rows[0]["Start_Timestamp"])
prev_end = t0
for r in rows:
    nm = r["Kernel_Name"]
    m = re.search(r"(k_[a-z_0-9]+(<[^>]*>)?)", nm)
    short = m.group(1) if m else re.sub(r"void |at::native::|\(anonymous namespace\)::", "", nm)[:70]
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%9.1f gap=%5.1f dur=%6.1f grid=%-8s %s" % ((s - t0) / 1e3, (s - prev_end) / 1e3, (e - s) / 1e3, r.get("Grid_Size") or r.get("Grid_Size_X", "?"), short[:70]))
    prev_end = e
