"""Summary of the PMC passes of tools/lab/pmc_lab.sh / pmc_lab2.sh over tools/micro/spmm_lab.bin (medians over the
timed dispatches of each kernel; every counter is the _sum over its instances).  usage: pmc_lab_summary.py DIR..."""
import collections, csv, glob, sys
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for d in sys.argv[1:]:
    for f in glob.glob(d + "/**/*_counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")
            if "fill" in k:
                continue
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, dct in agg.items():
    m = {c: sorted(v)[len(v) // 2] for c, v in dct.items()}
    print("== %s" % k)
    cyc = m.get("GRBM_GUI_ACTIVE", 0) / 8
    if cyc:
        print("   kernel cycles (GRBM_GUI_ACTIVE / 8 XCDs)            %12.0f" % cyc)
    for c in sorted(m):
        print("   %-50s %14.0f" % (c, m[c]))
    if "TCP_TCC_READ_REQ_sum" in m and m["TCP_TCC_READ_REQ_sum"]:
        lat = m["TCP_TCC_READ_REQ_LATENCY_sum"] / m["TCP_TCC_READ_REQ_sum"]
        print("   -> L1->L2 read requests of 128 B: %.0f; average latency %.0f cycles" % (m["TCP_TCC_READ_REQ_sum"], lat))
        if cyc:
            print("   -> read requests in flight per CU (latency sum / cycles / 256): %.1f" % (m["TCP_TCC_READ_REQ_LATENCY_sum"] / cyc / 256))
    if "TCP_TCC_WRITE_REQ_LATENCY_sum" in m and cyc:
        print("   -> write requests (64 B) in flight per CU: %.1f" % (m["TCP_TCC_WRITE_REQ_LATENCY_sum"] / cyc / 256))
    if "TCC_EA0_RDREQ_sum" in m:
        print("   -> L2 -> fabric reads: %.1f MB (128-byte requests), writes %.1f MB" % (m["TCC_EA0_RDREQ_sum"] * 128 / 1e6, m.get("TCC_EA0_WRREQ_sum", 0) * 64 / 1e6))
