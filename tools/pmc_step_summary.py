"""Per-kernel sums of one rocprofv3 --pmc pass over tools/step_only.py: which kernels of the train step have LDS bank
conflicts, how busy the matrix pipe is.    python tools/pmc_step_summary.py DIR [top]"""
import csv, glob, re, sys, collections
top = int(sys.argv[2]) if len(sys.argv) > 2 else 30
acc = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        nm = r["Kernel_Name"]
        m = re.search(r"(k_[a-z_0-9]+(<[^>]*>)?)", nm)
        key = m.group(1) if m else re.sub(r"void |at::native::|\(anonymous namespace\)::", "", nm)[:60]
        acc[key][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "SQ_WAVES":
            cnt[key] += 1
rows = sorted(acc.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0.0))[:top]
print("%-46s %7s %12s %9s %9s %9s" % ("kernel", "calls", "wave cyc(M)", "lds act%", "conflict%", "mfma/wave%"))
for k, d in rows:
    wc = d.get("SQ_WAVE_CYCLES", 0.0) * 4          # quad-cycles -> cycles summed over waves
    act, conf, mf = d.get("SQ_LDS_IDX_ACTIVE", 0.0), d.get("SQ_LDS_BANK_CONFLICT", 0.0), d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
    print("%-46s %7d %12.1f %9.1f %9.1f %9.1f" % (k[:46], cnt[k], wc / 1e6, 100 * act / max(d.get("SQ_BUSY_CYCLES", 1.0), 1.0),
                                                   100 * conf / act if act else 0.0, 100 * mf / wc if wc else 0.0))
