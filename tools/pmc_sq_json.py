"""Per-kernel SQ counters from the two rocprofv3 --pmc passes of tools/collect_sq_counters.sh, as the stamped JSON bench.py reads
(`roofline.sq_counters`: vector instructions per MFMA, share of SIMD cycles the vector ALU is active).

    python tools/pmc_sq_json.py <pass-1 counter_collection.csv> <pass-2 counter_collection.csv> <out.json> <mode>

Mean counter value per launch, template arguments stripped from the kernel names (instantiations of one kernel are averaged over
their launches).  Merged into <out.json> under `mode` and stamped with the digest of the kernel sources of this tree, like
tools/pmc_traffic.py does for the traffic set."""
import collections
import csv
import json
import re
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent))
from pmc_traffic_digest import csrc_digest  # noqa: E402


def collect(path):
    tot = collections.defaultdict(lambda: collections.defaultdict(float))
    launches = collections.defaultdict(set)
    for r in csv.DictReader(open(path)):
        k = re.sub(r"^void ", "", r["Kernel_Name"].split("(")[0]).replace("m3g::", "")
        rev = re.match(r"k_threebody_moments<.*,\s*(true|false)>", k)
        k = re.sub(r"<.*", "", k) + (("_rev" if rev.group(1) == "true" else "_fwd") if rev else "")
        if not k.startswith("k_"):
            continue
        tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
        launches[k].add(r["Dispatch_Id"])
    return {k: {c: v / len(launches[k]) for c, v in cs.items()} for k, cs in tot.items()}


def main():
    p1, p2, out_path, mode = sys.argv[1], sys.argv[2], Path(sys.argv[3]), sys.argv[4]
    a, b = collect(p1), collect(p2)
    kernels = {k: dict(a.get(k, {}), **b.get(k, {})) for k in sorted(set(a) | set(b))}
    doc = json.loads(out_path.read_text()) if out_path.exists() else {}
    digest = csrc_digest()
    if doc.get("_source", {}).get("csrc_sha256") not in (None, digest):
        doc = {}
    doc["_source"] = {"csrc_sha256": digest, "git_commit": doc.get("_source", {}).get("git_commit"),
                      "command": "rocprofv3 --kernel-trace --pmc <SQ counters, two passes: tools/collect_sq_counters.sh> -- python3 bench.py "
                                 "--steps 3 --warmup 1 --no-cpu-baseline --no-secondary --precision <mode>",
                      "units": "SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_* count units of 4 cycles summed over SIMDs; SQ_BUSY_CU_CYCLES cycles summed over CUs"}
    doc[mode] = kernels
    out_path.write_text(json.dumps(doc, indent=1))
    for k in ("k_edge_block_mfma", "k_edge_rev_f32", "k_edge_rev_fused"):
        if k in kernels and kernels[k].get("SQ_INSTS_MFMA"):
            c = kernels[k]
            print(f"{mode} {k}: VALU {c['SQ_INSTS_VALU'] / 1e6:.1f} M, MFMA {c['SQ_INSTS_MFMA'] / 1e6:.2f} M per launch = "
                  f"{c['SQ_INSTS_VALU'] / c['SQ_INSTS_MFMA']:.2f} VALU per MFMA; vector ALU active {c['SQ_ACTIVE_INST_VALU'] / c['SQ_BUSY_CU_CYCLES']:.2f} "
                  f"of SIMD cycles, matrix pipe busy {c['SQ_VALU_MFMA_BUSY_CYCLES'] / (4 * c['SQ_BUSY_CU_CYCLES']):.2f}")


if __name__ == "__main__":
    main()
