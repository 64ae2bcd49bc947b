#!/usr/bin/env python3
"""Audit of the hand-managed data hazards around inline-assembly blocks in a gfx950 listing (hipcc -S).

hipcc's hazard recogniser pads nothing INSIDE an `asm` statement and, at its boundary, only the dst_sel forwarding and
12-dword store cases (LLVM GCNHazardRecognizer::checkInlineAsmHazards).  Every other software-managed hazard whose consumer
sits inside an asm block is the author's: for each vector instruction inside a `;;#ASMSTART ... ;;#ASMEND` block this tool
finds, per source VGPR, the last writer before it and checks the wait states in between against the gfx940/gfx950 table:

    VALU write VGPR        -> DPP read of that VGPR                      2
    VALU write EXEC        -> DPP instruction                            5
    transcendental write   -> VALU read                                  1
    dst_sel / op_sel write -> VALU read                                  1   (compiler pads this one at the boundary)
    MFMA (XDL) write VGPR  -> VALU read or write of an overlapping VGPR  passes + 3   (4-pass 7, 8-pass 11, 16-pass 19)

Wait states = instructions issued in between (an `s_nop N` counts N + 1).  Usage:
    python tools/asm_hazard_audit.py listing.s [kernel-name-substring ...]
Prints one line per asm block class with the minimum margin seen, and every violation.
"""
from __future__ import annotations

import re
import sys
from collections import defaultdict

TRANS = ("v_exp_", "v_log_", "v_rcp_", "v_rsq_", "v_sqrt_", "v_sin_", "v_cos_")
MFMA_PASSES = {"v_mfma_f32_16x16x32_bf16": 4, "v_mfma_f32_16x16x4_f32": 8, "v_mfma_f32_32x32x2_f32": 16, "v_mfma_f32_32x32x16_bf16": 8,
               "v_mfma_f32_16x16x32_f16": 4}
REG = re.compile(r"\b([va])(\d+)\b|\b([va])\[(\d+):(\d+)\]")


def regs(tok: str):
    out = []
    for m in REG.finditer(tok):
        if m.group(1):
            out.append((m.group(1), int(m.group(2))))
        else:
            out += [(m.group(3), i) for i in range(int(m.group(4)), int(m.group(5)) + 1)]
    return out


def parse(line: str):
    code = line.split(";")[0].strip()
    if not code or code.endswith(":") or code.startswith("."):
        return None
    parts = code.split(None, 1)
    op = parts[0]
    ops = [o.strip() for o in parts[1].split(",")] if len(parts) > 1 else []
    return op, ops


def audit_kernel(name, lines):
    last = {}          # (file, n) -> (index of writer in wait-state units, class, passes)
    exec_write = -10**9
    t = 0              # wait-state clock
    in_asm = False
    blocks = defaultdict(lambda: [0, 10**9])   # first op of block -> [count, min margin]
    violations = []
    block_key = None
    for raw in lines:
        if ";;#ASMSTART" in raw:
            in_asm, block_key = True, None
            continue
        if ";;#ASMEND" in raw:
            in_asm = False
            continue
        p = parse(raw)
        if p is None:
            continue
        op, ops = p
        if op == "s_nop":
            t += int(ops[0], 0) + 1
            continue
        is_vec = op.startswith("v_")
        is_dpp = is_vec and ("_dpp" in op or any(k in raw for k in ("row_shr", "row_shl", "quad_perm", "row_bcast", "row_ror", "wave_shr")))
        if in_asm and is_vec:
            if block_key is None:
                block_key = op
                blocks[block_key][0] += 1
            srcs = ops[1:] if not op.startswith("v_fmac") and not op.startswith("v_mfma") else ops  # fmac / mfma also read dst / srcC
            for k, tok in enumerate(srcs):
                for r in regs(tok.split(" ")[0]):
                    w = last.get(r)
                    if w is None:
                        continue
                    wt, cls, passes = w
                    need = 0
                    if cls == "mfma":
                        need = passes + 3
                    elif cls == "trans":
                        need = 1
                    elif cls == "valu" and is_dpp and tok is srcs[1 if op.startswith("v_fmac") else 0]:
                        need = 2   # the DPP-routed operand (src0)
                    if is_dpp and cls == "valu":
                        need = max(need, 2 if tok is srcs[1 if op.startswith("v_fmac") else 0] else 0)
                    margin = (t - wt - 1) - need
                    if need:
                        blocks[block_key][1] = min(blocks[block_key][1], margin)
                        if margin < 0:
                            violations.append(f"{name}: {op} reads {r[0]}{r[1]} written by a {cls} op {t - wt - 1} wait states earlier, needs {need}")
            if is_dpp:
                margin = (t - exec_write - 1) - 5
                blocks[block_key][1] = min(blocks[block_key][1], margin)
                if margin < 0:
                    violations.append(f"{name}: {op} follows a VALU write of EXEC by {t - exec_write - 1} wait states, needs 5")
        # record writes
        if is_vec and ops:
            cls = "mfma" if op.startswith("v_mfma") else "trans" if op.startswith(TRANS) else "valu"
            passes = MFMA_PASSES.get(op.split("_e64")[0], 8) if cls == "mfma" else 0
            for r in regs(ops[0].split(" ")[0]):
                last[r] = (t, cls, passes)
            if op.startswith("v_cmpx"):
                exec_write = t
            if op.startswith("v_pk_") or "op_sel" in raw:
                pass   # dst_sel forwarding is padded by the compiler at the asm boundary
        elif op.startswith(("global_load", "ds_read", "buffer_load", "scratch_load", "flat_load")) and ops:
            for r in regs(ops[0]):
                last.pop(r, None)   # memory returns are ordered by s_waitcnt, not by wait states
        t += 1
    return blocks, violations


def main():
    path = sys.argv[1]
    filters = sys.argv[2:]
    kernels, cur, name = {}, None, None
    for line in open(path, errors="replace"):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            name, cur = m.group(1), []
            kernels[name] = cur
            continue
        if cur is not None:
            cur.append(line)
            if "s_endpgm" in line:
                cur = None
    total_v = 0
    for name, lines in kernels.items():
        if filters and not any(f in name for f in filters):
            continue
        blocks, violations = audit_kernel(name, lines)
        if not blocks:
            continue
        desc = ", ".join(f"{n} x asm[{op}] min margin {mm if mm < 10**8 else 'n/a'}" for op, (n, mm) in blocks.items())
        print(f"{name[:70]}: {desc}; violations {len(violations)}")
        for v in violations[:10]:
            print("   ", v)
        total_v += len(violations)
    print("total violations:", total_v)
    return 1 if total_v else 0


if __name__ == "__main__":
    sys.exit(main())
