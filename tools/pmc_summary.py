"""Per-kernel mean of each counter from a rocprofv3 --pmc CSV directory."""
import csv, glob, sys, collections, re
d = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        m = re.search(r"(k_[a-z_]+(<[^>]*>)?)", r["Kernel_Name"])
        if not m:
            continue
        acc[m.group(1)][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in sorted(acc.items()):
    print(k)
    print("   ", " ".join("%s=%.4g" % (c, sum(v) / len(v)) for c, v in sorted(cs.items())), "n=%d" % len(next(iter(cs.values()))))
