"""Per-step kernel statistics of a rocprofv3 --kernel-trace of tools/step_only.py: only the launches between the
two marker launches (the K timed steps) count, so calls / K is the true per-step launch count.

    python tools/step_trace_summary.py DIR K [top]     -> by kernel, then by (kernel, grid)
"""
import collections, csv, glob, re, sys
d, steps = sys.argv[1], float(sys.argv[2])
top = int(sys.argv[3]) if len(sys.argv) > 3 else 40
rows = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: float(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(rows) if re.search(r"cumsum|scan_innermost|ScanInnermost|tensor_kernel_scan|lookback_scan|single_scan|scan_kernel", r["Kernel_Name"])]
if len(marks) < 2:
    sys.exit("markers not found (%d)" % len(marks))
rows = rows[marks[0] + 1: marks[-1]]
wall = (float(rows[-1]["End_Timestamp"]) - float(rows[0]["Start_Timestamp"])) / 1e6


def short(nm):
    m = re.search(r"(k_[a-z_0-9]+(<[^>]*>)?)", nm)
    return m.group(1) if m else re.sub(r"void |at::native::|\(anonymous namespace\)::", "", nm)[:70]


by_k, by_s = collections.defaultdict(list), collections.defaultdict(list)
for r in rows:
    dt = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
    by_k[short(r["Kernel_Name"])].append(dt)
    by_s[(short(r["Kernel_Name"]), r.get("Grid_Size") or r.get("Grid_Size_X", "?"))].append(dt)
tot = sum(sum(v) for v in by_k.values())
print("between the markers: %d steps, %.3f ms/step wall (first launch start to last launch end), %.3f ms/step inside kernels, "
      "%.1f launches/step" % (steps, wall / steps, tot / 1e6 / steps, len(rows) / steps))
print("\n== by kernel: name, calls/step, avg us, ms/step, share")
for k, v in sorted(by_k.items(), key=lambda kv: -sum(kv[1]))[:top]:
    print("%-70s %7.2f %8.2f %8.4f %6.2f%%" % (k[:70], len(v) / steps, sum(v) / len(v) / 1e3, sum(v) / 1e6 / steps, 100 * sum(v) / tot))
print("\n== by kernel and grid: name, grid, calls/step, avg us, ms/step")
for (k, g), v in sorted(by_s.items(), key=lambda kv: -sum(kv[1]))[:top]:
    print("%-62s grid=%-9s %7.2f %8.2f %8.4f" % (k[:62], g, len(v) / steps, sum(v) / len(v) / 1e3, sum(v) / 1e6 / steps))
