#!/usr/bin/env python3
"""Build DESIGN.md = docs/DESIGN_part1.md (the design as it stands, numbers taken from the committed bench line) + the measurement
history of earlier rounds (docs/DESIGN_history_r1_r3.md, the previous DESIGN.md with its headings marked II.).

    python tools/assemble_design.py profiles/r04_bench_default_run.json
"""
import json
import re
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
line = [l for l in Path(sys.argv[1]).read_text().splitlines() if l.startswith("{")][-1]
d = json.loads(line)
pm = json.loads((ROOT / "profiles" / "r04_pmc_hbm_traffic.json").read_text())


def step_gb(mode):
    return sum(r["launches_per_step"] * (2 * r["fetch_kb"] + r["write_kb"]) * 1024 for r in pm[mode].values() if r["launches_per_step"] >= 0.99) / 1e9


md = d["beside"]["md_iteration_ms_10k_atom_cell"]
f16, bf16 = d["f16x3"], d["bf16x3"]
vals = {
    "FP32_MS": f"{d['ms_per_step']:.3f}", "FP32_VALUE": f"{d['value'] / 1e6:.2f}", "FP32_MIN": f"{d['ms_per_step_min']:.3f}",
    "FP32_MED": f"{d['ms_per_step_median']:.3f}", "FP32_CLK": f"{d['clock_mhz']:.0f}" if d.get("clock_mhz") else "n/a",
    "F16_MS": f"{f16['ms_per_step']:.3f}", "BF16_MS": f"{bf16['ms_per_step']:.3f}",
    "CPU_MS": f"{d['cpu_baseline']['ms_per_step']:.0f}", "CPU_VALUE": f"{d['cpu_baseline']['value']:,.0f}",
    "MD_REUSE": f"{md['fp32']['reuse_verdict_read_after_the_step']['total']:.2f}", "MD_REUSE_F16": f"{md['f16x3']['reuse_verdict_read_after_the_step']['total']:.2f}",
    "MD_REBUILD": f"{md['fp32']['rebuild']['total']:.2f}", "MD_REBUILD_F16": f"{md['f16x3']['rebuild']['total']:.2f}",
    "MD_REFILL": f"{md['fp32']['refill']['total']:.2f}", "MD_REFILL_F16": f"{md['f16x3']['refill']['total']:.2f}",
    "FP32_GB": f"{step_gb('fp32'):.2f}", "F16_GB": f"{step_gb('f16x3'):.2f}", "BF16_GB": f"{step_gb('bf16x3'):.2f}",
    "FP32_REV_US": f"{d['roofline']['avg_launch_ms'] * 1e3:.0f}", "FP32_FRAC": f"{d['roofline']['frac']:.2f}",
    "F16_REV_US": f"{f16['roofline']['avg_launch_ms'] * 1e3:.0f}", "F16_FRAC": f"{f16['roofline']['frac']:.2f}",
    "F16_HBM": f"{f16['roofline']['hbm_traffic_view']['frac']:.2f}",
    "C4_MS": f"{d['config4_sharded']['ms_per_step']:.2f}", "SMALL_MS": f"{d['beside']['step_ms_32_atom_cu_cell']:.3f}",
}
part1 = (ROOT / "docs" / "DESIGN_part1.md").read_text()
missing = set(re.findall(r"@([A-Z0-9_]+)@", part1)) - set(vals)
assert not missing, missing
for k, v in vals.items():
    part1 = part1.replace(f"@{k}@", v)
history = (ROOT / "docs" / "DESIGN_history_r1_r3.md").read_text()
(ROOT / "DESIGN.md").write_text(part1 + history)
print("DESIGN.md written:", len(part1.splitlines()), "+", len(history.splitlines()), "lines")
