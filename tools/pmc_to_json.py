"""HBM traffic per launch of the roofline kernels from two rocprofv3 --pmc passes (FETCH_SIZE; WRITE_SIZE)
over tools/roofline_kernels.py -> JSON read by bench.py.  FETCH_SIZE is doubled (gfx950 tallies 128-byte
requests at 64 bytes, MI355X_MICROARCH.md); both counters are in KiB."""
import csv, glob, json, sys, collections
fetch_dir, write_dir, out = sys.argv[1], sys.argv[2], sys.argv[3]
# (every substring of a key's tuple must occur in the kernel name)
KEYS = {"gemm_generator_block": ("k_gemm_panel<", "false>"), "gemm_generator_block_fused": ("k_gemm_panel<", "true>"),
        "spmm_H128": ("k_spmm_batched<32,",), "spmm_H256": ("k_spmm_batched<64,",)}
def mean(d, counter):
    acc = collections.defaultdict(list)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            for k, pat in KEYS.items():
                if all(p_ in r["Kernel_Name"] for p_ in pat):
                    acc[k].append(float(r["Counter_Value"]))
    return {k: sum(v[2:]) / len(v[2:]) for k, v in acc.items() if len(v) > 2}
fe, wr = mean(fetch_dir, "FETCH_SIZE"), mean(write_dir, "WRITE_SIZE")
res = {k: round((2 * fe[k] + wr.get(k, 0.0)) * 1024) for k in fe}
res["_fetch_KiB"], res["_write_KiB"] = fe, wr
res["_note"] = ("bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) KiB, separate --pmc passes over tools/roofline_kernels.py "
                "(gather-mean: rotating over 4 operand / result pairs, as bench.py's microbench)")
json.dump(res, open(out, "w"), indent=1)
print(res)
