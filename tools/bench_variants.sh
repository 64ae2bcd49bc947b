#!/bin/bash
# usage: [PREC=f16x3] [STEPS=30] [BENCH_ARGS="--engine-option threebody_moments=0"] tools/bench_variants.sh name1 name2 ...   (runs bench.py with lib/variants/<name>.so swapped in; "base" = the built library)
cd "$(dirname "$0")/.."
L=torch-m3gnet_amd/lib
cp $L/libm3gnet_hip.so /tmp/base.so
for v in "$@"; do
  if [ "$v" = base ]; then cp /tmp/base.so $L/libm3gnet_hip.so; else cp $L/variants/$v.so $L/libm3gnet_hip.so; fi
  timeout -k 10 200 python bench.py --steps ${STEPS:-10} --warmup 3 --no-cpu-baseline --no-secondary --precision ${PREC:-fp32} $BENCH_ARGS > gpurun_out/bench_var_$v.json 2> gpurun_out/bench_var_$v.err || { echo "$v FAILED"; tail -3 gpurun_out/bench_var_$v.err; continue; }
  python - "$v" <<PY
import json,sys
d=json.loads(open("gpurun_out/bench_var_%s.json"%sys.argv[1]).read().strip().splitlines()[-1])
s=d["config"]["stage_ms_per_step"]
print(sys.argv[1], "ms/step %.4f"%d["ms_per_step"], {k: s[k] for k in ("edge_block_fwd","edge_rev_fused","edge_rev_node_mlp","edge_rev_edge_mlp","node_rev","node_pre","threebody_rev","threebody_fwd") if k in s})
PY
done
cp /tmp/base.so $L/libm3gnet_hip.so
