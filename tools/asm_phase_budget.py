#!/usr/bin/env python3
"""Per-phase instruction budget of a STAMPED kernel variant from its ISA listing: the stamped builds read s_memtime at every phase
boundary of the tile loop (Stamps<true>::start / mark<k>), so the instructions between two consecutive s_memtime reads in the
listing are the phase's static instruction stream.  Counts by class (MFMA, plain / packed / transcendental vector, LDS, vector
memory, scalar) per phase, in listing order.

    hipcc -O3 -std=c++20 --offload-arch=gfx950 -Iinclude -S --cuda-device-only -o /tmp/edge.s torch-m3gnet_amd/csrc/m3g_edge_mfma.hip
    python tools/asm_phase_budget.py /tmp/edge.s k_edge_rev_fusedILi3ELb1ELi8ELi2ELb1E "n: inputs, tables, layer 1" ...   (phase names, optional)
"""
import collections
import re
import sys

from asm_mix import classify


def main():
    path, key = sys.argv[1], sys.argv[2]
    names = sys.argv[3:]
    lines = open(path).read().splitlines()
    start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w*" + re.escape(key) + r"\w*:", l))
    end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
    segs, cur = [], collections.Counter()
    for l in lines[start:end]:
        t = l.strip()
        if not t or t.startswith((";", ".", "//")) or t.endswith(":"):
            continue
        op = t.split()[0]
        if op == "s_memtime":
            segs.append(cur)
            cur = collections.Counter()
            continue
        cur[classify(op)] += 1
        if op.startswith("v_") and "_dpp" in op:
            cur["dpp"] += 1
    segs.append(cur)
    cols = ["mfma", "valu", "v_pk", "trans", "lds", "vmem", "salu"]
    print(f"{'segment':44s} " + " ".join(f"{c:>6s}" for c in cols) + "   vector total")
    tot = collections.Counter()
    for i, s in enumerate(segs):
        # segment 0 = everything before the first stamp (prologue + loop head), the last one = after the last stamp (loop tail, epilogue)
        label = "(before the first stamp: prologue)" if i == 0 else (names[i - 1] if i - 1 < len(names) else f"segment {i}")
        vec = s["valu"] + s["v_pk"] + s["trans"]
        print(f"{label[:44]:44s} " + " ".join(f"{s[c]:6d}" for c in cols) + f"   {vec:6d}")
        if i > 0:
            tot.update(s)
    vec = tot["valu"] + tot["v_pk"] + tot["trans"]
    print(f"{'tile loop (segments after the first stamp)':44s} " + " ".join(f"{tot[c]:6d}" for c in cols) + f"   {vec:6d}")


if __name__ == "__main__":
    main()
