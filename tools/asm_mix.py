#!/usr/bin/env python3
"""Instruction mix per kernel of a `hipcc -S --cuda-device-only` listing: counts by class (MFMA, packed VALU, transcendental,
other VALU, LDS, VMEM, SALU) for kernels whose mangled name contains one of the given substrings.

    hipcc -O3 -std=c++20 --offload-arch=gfx950 -Iinclude -S --cuda-device-only -o /tmp/edge.s torch-m3gnet_amd/csrc/m3g_edge_mfma.hip
    python tools/asm_mix.py /tmp/edge.s rev_fusedILi3ELb1 edge_block_mfmaILi3ELb1
"""
import collections
import re
import sys


def classify(op):
    if op.startswith("v_mfma"):
        return "mfma"
    if op.startswith("v_pk_"):
        return "v_pk"
    if op.startswith(("v_exp", "v_rcp", "v_log", "v_sqrt", "v_rsq", "v_sin", "v_cos")):
        return "trans"
    if op.startswith("v_"):
        return "valu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    if op.startswith("s_"):
        return "salu"
    return "other"


def main():
    path, pats = sys.argv[1], sys.argv[2:]
    name, ops = None, None
    out = []
    for line in open(path):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            if name:
                out.append((name, ops))
            name, ops = m.group(1), collections.Counter()
            continue
        if name is None:
            continue
        if line.startswith(".Lfunc_end"):
            out.append((name, ops))
            name = None
            continue
        m = re.match(r"^\s+([a-z][a-z_0-9]+)\s", line)
        if m:
            ops[m.group(1)] += 1
    for name, ops in out:
        if pats and not any(p in name for p in pats):
            continue
        cls = collections.Counter()
        for k, v in ops.items():
            cls[classify(k)] += v
        print(name[:90], "total", sum(ops.values()))
        print("  ", dict(cls))
        print("  ", [(k, v) for k, v in ops.most_common(28)])


if __name__ == "__main__":
    main()
