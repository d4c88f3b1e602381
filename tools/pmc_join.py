"""Join a rocprofv3 --pmc + --kernel-trace run: per kernel name, mean duration and counters."""
import csv, glob, sys, collections, re
d = sys.argv[1]
pat = sys.argv[2] if len(sys.argv) > 2 else "k_gemm"
dur = {}
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        dur[r["Dispatch_Id"]] = (r["Kernel_Name"], float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if pat not in r["Kernel_Name"]:
            continue
        m = re.search(r"(k_[a-z_]+(<[^>]*>)?)", r["Kernel_Name"])
        key = (m.group(1), r["Grid_Size"])
        acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
        if r["Dispatch_Id"] in dur:
            acc[key]["_dur_ns"].append(dur[r["Dispatch_Id"]][1])
for k, cs in sorted(acc.items()):
    m = {c: sum(v) / len(v) for c, v in cs.items()}
    line = "%s grid=%s dur_us=%.1f" % (k[0], k[1], m.get("_dur_ns", 0) / 1e3)
    if "GRBM_GUI_ACTIVE" in m and m.get("_dur_ns"):
        cyc = m["GRBM_GUI_ACTIVE"] / 8
        line += " clk_GHz=%.2f" % (cyc / m["_dur_ns"])
        if "SQ_VALU_MFMA_BUSY_CYCLES" in m:
            line += " mfma_util=%.3f" % (m["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * cyc))
    print(line, {c: "%.3g" % v for c, v in m.items() if not c.startswith("_")})
