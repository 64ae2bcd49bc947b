"""Per-kernel HBM traffic from two rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE cannot share a pass on gfx950:
MI355X_MICROARCH.md, counter table).  Mean KiB per launch and launches per step (relative to k_geometry, which runs once per
step), template arguments stripped from kernel names.

    python tools/pmc_traffic.py <fetch counter_collection.csv> <write counter_collection.csv> <out.json> [mode [command]]

With `mode` (fp32 | bf16x3) the result is MERGED into <out.json> under that key and the file is stamped with the digest of the
kernel sources of this tree (`_source.csrc_sha256`, the same digest bench.py computes at run time: a profile set from other
sources yields `traffic: null` in the bench line).  `_source.git_commit` is filled in by tools/stamp_profile_commit.py in the
build container (the GPU box has no .git).  bench.py corrects FETCH_SIZE by 2x (gfx950 tallies 128-byte requests at 64 B)."""
import collections
import csv
import hashlib
import json
import re
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent


def csrc_digest():
    h = hashlib.sha256()
    files = sorted((ROOT / "torch-m3gnet_amd" / "csrc").glob("*.h*")) + sorted((ROOT / "include").glob("*.h"))
    for f in files:
        h.update(f.name.encode())
        h.update(f.read_bytes())
    return h.hexdigest()[:16]


def collect(path, counter):
    tot, launches = collections.defaultdict(float), collections.defaultdict(set)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        k = re.sub(r"^void ", "", r["Kernel_Name"].split("(")[0]).replace("m3g::", "")
        rev = re.match(r"k_threebody_moments<.*,\s*(true|false)>", k)   # forward and reverse are one template: keep them apart
        k = re.sub(r"<.*", "", k) + (("_rev" if rev.group(1) == "true" else "_fwd") if rev else "")
        tot[k] += float(r["Counter_Value"])
        launches[k].add(r["Dispatch_Id"])
    return {k: tot[k] / len(launches[k]) for k in tot}, {k: len(v) for k, v in launches.items()}


(fetch, n_f), (write, n_w) = collect(sys.argv[1], "FETCH_SIZE"), collect(sys.argv[2], "WRITE_SIZE")
steps = max(1, n_f.get("k_geometry", 1))
kernels = {k: {"fetch_kb": fetch.get(k, 0.0), "write_kb": write.get(k, 0.0), "launches_per_step": n_f.get(k, n_w.get(k, 0)) / steps}
           for k in sorted(set(fetch) | set(write)) if k.startswith("k_")}
out_path = Path(sys.argv[3])
if len(sys.argv) > 4:
    doc = json.loads(out_path.read_text()) if out_path.exists() else {}
    digest = csrc_digest()
    if doc.get("_source", {}).get("csrc_sha256") not in (None, digest):
        doc = {}   # an older set from other sources: start over
    doc["_source"] = {"csrc_sha256": digest, "git_commit": doc.get("_source", {}).get("git_commit"),
                      "command": sys.argv[5] if len(sys.argv) > 5 else "rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE (separate passes) -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --precision <mode>"}
    doc[sys.argv[4]] = kernels
    out_path.write_text(json.dumps(doc, indent=1))
else:
    out_path.write_text(json.dumps(kernels, indent=1))
total = 0.0
for k, v in sorted(kernels.items(), key=lambda kv: -(2 * kv[1]["fetch_kb"] + kv[1]["write_kb"]) * kv[1]["launches_per_step"]):
    b = (2 * v["fetch_kb"] + v["write_kb"]) * 1024.0
    total += b * v["launches_per_step"] if v["launches_per_step"] >= 0.99 else 0.0   # (< 1: topology build of the first call)
    print(f"{k:32s} fetch {v['fetch_kb'] / 1024:9.1f} MiB (x2 corrected {2 * v['fetch_kb'] / 1024:9.1f})  write {v['write_kb'] / 1024:9.1f} MiB"
          f"   x {v['launches_per_step']:.2f} per step = {b * v['launches_per_step'] / 1e6:9.1f} MB")
print(f"{'whole step':32s} {total / 1e9:.3f} GB")
