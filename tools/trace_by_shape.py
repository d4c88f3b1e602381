"""Group a rocprofv3 --kernel-trace CSV by (kernel, grid size): calls, mean us, total ms (tuning aid)."""
import csv, glob, sys, collections, re
d = sys.argv[1]; steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0; top = int(sys.argv[3]) if len(sys.argv) > 3 else 40
acc = collections.defaultdict(list)
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        nm = r["Kernel_Name"]
        m = re.search(r"(k_[a-z_0-9]+(<[^>]*>)?)", nm)
        short = m.group(1) if m else re.sub(r"void |at::native::|\(anonymous namespace\)::", "", nm)[:60]
        acc[(short, r.get("Grid_Size") or r.get("Grid_Size_X", "?"))].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
tot = sum(sum(v) for v in acc.values())
print("total %.3f ms/step, %.0f launches/step" % (tot / 1e6 / steps, sum(len(v) for v in acc.values()) / steps))
for (k, g), v in sorted(acc.items(), key=lambda kv: -sum(kv[1]))[:top]:
    print("%-62s grid=%-9s n/step=%6.1f avg=%7.1f us  %6.3f ms/step" % (k[:62], g, len(v) / steps, sum(v) / len(v) / 1e3, sum(v) / 1e6 / steps))
