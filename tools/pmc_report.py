"""Aggregate a rocprofv3 --pmc counter_collection.csv per kernel: mean counter value per launch."""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
filt = sys.argv[2] if len(sys.argv) > 2 else ""
agg = collections.defaultdict(lambda: collections.defaultdict(float))
launches = collections.defaultdict(set)
for r in rows:
    k = r["Kernel_Name"].split("(")[0][-48:]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    launches[k].add(r["Dispatch_Id"])
for k, v in agg.items():
    if filt in k:
        n = len(launches[k])
        print(k, f"launches={n}", {c: round(x / n) for c, x in v.items()})
