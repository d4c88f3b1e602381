"""Group a kernel trace into runs of consecutive identical (kernel, grid) launches; mean us of each run."""
import csv, glob, sys, re
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
runs = []
for r in rows:
    m = re.search(r"(k_[a-z_0-9]+(<[^>]*>)?)", r["Kernel_Name"])
    if not m or "k_gemm" not in m.group(1):
        continue
    key = (m.group(1), r.get("Grid_Size") or r.get("Grid_Size_X"))
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    if runs and runs[-1][0] == key:
        runs[-1][1].append(d)
    else:
        runs.append((key, [d]))
for key, ds in runs:
    ds = ds[2:] if len(ds) > 4 else ds
    print("%-48s grid=%-8s n=%2d  %6.1f us" % (key[0], key[1], len(ds), sum(ds) / len(ds)))
