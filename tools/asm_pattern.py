#!/usr/bin/env python3
"""Instruction-class pattern of each kernel in a `hipcc -S` listing: M = MFMA, T = transcendental, V = other VALU, D = LDS,
G = global/buffer memory, W = s_waitcnt, N = s_nop, s = other scalar.  usage: asm_pattern.py listing.s [name-substring] [width]"""
import re
import sys

txt = open(sys.argv[1]).read()
want = sys.argv[2] if len(sys.argv) > 2 else ""
width = int(sys.argv[3]) if len(sys.argv) > 3 else 400
parts = re.split(r"\n(_Z\w+):[^\n]*\n", txt)
for i in range(1, len(parts), 2):
    name, body = parts[i], parts[i + 1].split(".Lfunc_end")[0]
    if want not in name:
        continue
    lines = [l.strip() for l in body.splitlines() if l.strip() and not l.strip().startswith((".", ";"))]
    lines = [l for l in lines if not l.endswith(":")]

    def cls(l):
        if l.startswith("v_mfma"):
            return "M"
        if re.match(r"v_(exp|rcp|log|rsq|sqrt|sin|cos)", l):
            return "T"
        if l.startswith("v_"):
            return "V"
        if l.startswith("ds_"):
            return "D"
        if l.startswith(("global_", "buffer_", "flat_", "scratch_")):
            return "G"
        if l.startswith("s_waitcnt"):
            return "W"
        if l.startswith("s_nop"):
            return "N"
        return "s"

    seq = "".join(cls(l) for l in lines)
    counts = {c: seq.count(c) for c in "MTVDGWNs"}
    print(name, len(lines), counts)
    k = seq.find("M")
    print(seq[k:k + width])
    # gaps between MFMAs: histogram of non-MFMA instructions per gap
    gaps = [len(g) for g in seq[k:].split("M")[1:-1]]
    hist = {}
    for g in gaps:
        hist[g] = hist.get(g, 0) + 1
    print("gap histogram (instructions between consecutive MFMAs):", dict(sorted(hist.items())))
