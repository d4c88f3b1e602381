"""Per-kernel means of the SQ counters of one rocprofv3 --pmc pass (tools/roofline_kernels.py): MFMA busy share etc.
    python tools/pmc_sq_summary.py DIR"""
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        nm = r["Kernel_Name"]
        key = "k_gemm_panel" if "k_gemm_panel<" in nm else ("k_spmm_batched<32" if "k_spmm_batched<32," in nm else
              ("k_spmm_batched<64" if "k_spmm_batched<64," in nm else None))
        if key:
            acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    m = {c: sum(v[2:]) / max(1, len(v[2:])) for c, v in d.items()}
    print(k, {c: round(x, 1) for c, x in m.items()})
    if "SQ_VALU_MFMA_BUSY_CYCLES" in m and "SQ_BUSY_CYCLES" in m and m["SQ_BUSY_CYCLES"] > 0:
        print("   MFMA busy / SQ busy cycles = %.3f" % (m["SQ_VALU_MFMA_BUSY_CYCLES"] / m["SQ_BUSY_CYCLES"]))
    if "SQ_LDS_BANK_CONFLICT" in m and "SQ_LDS_IDX_ACTIVE" in m and m["SQ_LDS_IDX_ACTIVE"] > 0:
        print("   LDS bank-conflict cycles / LDS active cycles = %.4f" % (m["SQ_LDS_BANK_CONFLICT"] / m["SQ_LDS_IDX_ACTIVE"]))
