#!/usr/bin/env python3
"""Kernel sequence of ONE step from a `rocprofv3 --kernel-trace --output-format csv` run: every dispatch between the last two
k_geometry<> launches, with its duration and the gap to the previous dispatch's end (us).

    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/seq -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary
    python tools/step_sequence.py gpurun_out/seq [k [marker]]     (marker: substring of the kernel that starts an iteration, default
                                                          k_geometry<; k: which step, counted from the end; default 1 = the last one.  bench.py
                                                          ends with `--steps` PROFILED steps -- stage events between the kernels --,
                                                          so k = steps + 1 is the last step of the timed region)
"""
import csv
import glob
import sys


def main():
    paths = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)
    rows = []
    for p in paths:
        rows += list(csv.DictReader(open(p)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    # a kernel that runs once per iteration, first: the geometry stage (small systems: fused with block 0's node tables)
    markers = [sys.argv[3]] if len(sys.argv) > 3 else ["k_geometry<", "k_geometry_node_pre<"]
    geo = [i for i, r in enumerate(rows) if any(m in r["Kernel_Name"] for m in markers)]
    if len(geo) < 2:
        print("fewer than two steps in the trace")
        return
    k = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    if len(geo) < k + 1:
        print("fewer steps in the trace than asked for")
        return
    a, b = geo[-k - 1], geo[-k]
    prev_end = None
    total = 0.0
    for r in rows[a:b]:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        gap = (s - prev_end) / 1e3 if prev_end is not None else 0.0
        print(f"{(e - s) / 1e3:9.1f} us  gap {gap:7.1f}  {r['Kernel_Name'][:100]}")
        total += (e - s) / 1e3
        prev_end = e
    span = (int(rows[b]["Start_Timestamp"]) - int(rows[a]["Start_Timestamp"])) / 1e3
    print(f"step span {span:.1f} us, sum of kernel durations {total:.1f} us, {b - a} dispatches")


if __name__ == "__main__":
    main()
