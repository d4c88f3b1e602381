"""Launch-ordered kernel list of ONE step from a rocprofv3 --kernel-trace of tools/step_only.py (tuning aid).
    python tools/step_trace_order.py DIR K which_step"""
import csv, glob, re, sys
d, steps, which = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
rows = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: float(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(rows) if re.search(r"cumsum|scan_innermost|tensor_kernel_scan|lookback_scan|single_scan|scan_kernel", r["Kernel_Name"])]
rows = rows[marks[0] + 1: marks[-1]]
# steps are delimited by the Adam launch of the critic (k_adam ends every iteration)
ends = [i for i, r in enumerate(rows) if "k_adam" in r["Kernel_Name"]]
per = len(ends) // steps
lo = ends[which * per - 1] + 1 if which > 0 else 0
hi = ends[(which + 1) * per - 1] + 1
t0 = float(rows[lo]["Start_Timestamp"])
for r in rows[lo:hi]:
    nm = r["Kernel_Name"]
    m = re.search(r"(k_[a-z_0-9]+(<[^>]*>)?)", nm)
    short = m.group(1) if m else re.sub(r"void |at::native::|\(anonymous namespace\)::", "", nm)[:70]
    print("%9.1f %7.1f  %-60s grid=%s wg=%s" % ((float(r["Start_Timestamp"]) - t0) / 1e3,
          (float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e3, short[:60],
          r.get("Grid_Size") or r.get("Grid_Size_X"), r.get("Workgroup_Size") or r.get("Workgroup_Size_X")))
