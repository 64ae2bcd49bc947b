"""Per-kernel HBM traffic from two rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE cannot share a pass on gfx950:
MI355X_MICROARCH.md, counter table).  Mean KiB per launch, template arguments stripped from kernel names.

    python tools/pmc_traffic.py <fetch counter_collection.csv> <write counter_collection.csv> <out.json>

bench.py corrects FETCH_SIZE by 2x (gfx950 tallies 128-byte requests at 64 B) when it prices `roofline.traffic`."""
import collections
import csv
import json
import re
import sys


def collect(path, counter):
    tot, launches = collections.defaultdict(float), collections.defaultdict(set)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        k = re.sub(r"^void ", "", r["Kernel_Name"].split("(")[0]).replace("m3g::", "")
        k = re.sub(r"<.*", "", k)
        tot[k] += float(r["Counter_Value"])
        launches[k].add(r["Dispatch_Id"])
    return {k: tot[k] / len(launches[k]) for k in tot}


fetch, write = collect(sys.argv[1], "FETCH_SIZE"), collect(sys.argv[2], "WRITE_SIZE")
out = {k: {"fetch_kb": fetch.get(k, 0.0), "write_kb": write.get(k, 0.0)} for k in sorted(set(fetch) | set(write)) if "rocprim" not in k}
json.dump(out, open(sys.argv[3], "w"), indent=1)
for k, v in sorted(out.items(), key=lambda kv: -(2 * kv[1]["fetch_kb"] + kv[1]["write_kb"])):
    print(f"{k:32s} fetch {v['fetch_kb'] / 1024:9.1f} MiB (x2 corrected {2 * v['fetch_kb'] / 1024:9.1f})  write {v['write_kb'] / 1024:9.1f} MiB")
