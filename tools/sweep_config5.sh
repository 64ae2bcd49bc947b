#!/bin/bash
# usage (GPU box, repo root; variants built by tools/build_variant_tb.sh): tools/sweep_config5.sh tb64 tb96 tb128 ...
# -> gpurun_out/config5_sweep.jsonl: one line per (rows per workgroup, three-body cutoff)
cd "$(dirname "$0")/.."
L=torch-m3gnet_amd/lib
cp $L/libm3gnet_hip.so /tmp/base.so
: > gpurun_out/config5_sweep.jsonl
for r in "$@"; do
  cp $L/variants/$r.so $L/libm3gnet_hip.so
  timeout -k 10 200 python tools/time_config5.py $r >> gpurun_out/config5_sweep.jsonl 2> gpurun_out/config5_sweep_$r.err || echo "rows $r FAILED"
done
cp /tmp/base.so $L/libm3gnet_hip.so
python - <<PY
import json
for line in open("gpurun_out/config5_sweep.jsonl"):
    d = json.loads(line)
    f, r = d["threebody_fwd"], d["threebody_rev"]
    print(f'{d["label"]:8s} r3={d["threebody_cutoff"]} T/atom={d["triplets_per_atom"]:.0f} step {d["ms_per_step"]:.3f} ms  fwd {f["ms_per_launch"]*1e3:.1f} us '
          f'(HBM {f["hbm_GBs"]:.0f} GB/s, LDS {f["lds_read_TBs"]:.1f} TB/s)  rev {r["ms_per_launch"]*1e3:.1f} us (HBM {r["hbm_GBs"]:.0f} GB/s, LDS {r["lds_read_TBs"]:.1f} TB/s)')
PY
