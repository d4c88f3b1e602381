"""Summarise a rocprofv3 --kernel-trace --stats CSV directory (tuning aid)."""
import csv, glob, sys
d = sys.argv[1]; steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
f = glob.glob(d + "/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("WHOLE bench command (warm-up, timed steps, hipGraph replay figure, roofline / microbench / inference kernels): "
      "total kernel ms %.2f, %d launches.  NOT a per-step figure -- the timed steps alone are in *_step_kernels.txt"
      % (tot / 1e6, sum(int(r["Calls"]) for r in rows)))
for r in rows[: int(sys.argv[3]) if len(sys.argv) > 3 else 22]:
    print("%-78s %7s %9.2f %8.1f %6s" % (r["Name"][:78], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3, r["Percentage"][:5]))
