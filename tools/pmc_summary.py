"""Summarise a rocprofv3 counter_collection.csv: mean counter value per dispatch for the ofdm:: kernels."""
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for path in sys.argv[1:]:
    with open(path) as fh:
        for row in csv.DictReader(fh):
            name = row.get("Kernel_Name", "")
            if "ofdm::" not in name:
                continue
            name = name.replace("(anonymous namespace)::", "")   # kernels_mid.hip keeps its kernels in an unnamed namespace
            acc[name.split("(")[0]][row["Counter_Name"]].append((int(row.get("Dispatch_Id", 0) or 0), float(row["Counter_Value"])))
for k, d in sorted(acc.items()):
    for c, pairs in sorted(d.items()):
        vals = [v for _, v in sorted(pairs)]
        # (per-call values in dispatch order: a kernel name may cover launches of different shapes, e.g. k_read_probe's three patterns)
        print(f"{k}\t{c}\tcalls={len(vals)}\tmean={sum(vals)/len(vals):.1f}\tmax={max(vals):.1f}\tvals={','.join(f'{v:.0f}' for v in vals[:24])}")
