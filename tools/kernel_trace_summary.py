"""Per-kernel totals of a rocprofv3 `--kernel-trace` results .db (rocpd sqlite): calls, total and mean duration, sorted by total.
Usage: python tools/kernel_trace_summary.py results.db [divide_by]   (divide_by: e.g. the number of iterations traced)"""
import collections
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
div = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
tot, cnt = collections.Counter(), collections.Counter()
for name, start, end in db.execute("select name, start, end from kernels"):
    short = re.sub(r"\(.*", "", name)
    short = re.sub(r"^(void )?", "", short)[:70]
    tot[short] += end - start
    cnt[short] += 1
grand = sum(tot.values())
print(f"{'kernel':70s} {'calls':>8s} {'total us':>10s} {'mean us':>9s}  (per 1/{div:g})")
for k, v in tot.most_common():
    print(f"{k:70s} {cnt[k] / div:8.1f} {v / 1e3 / div:10.1f} {v / 1e3 / cnt[k]:9.1f}")
print(f"{'sum':70s} {'':8s} {grand / 1e3 / div:10.1f}")
