#!/bin/bash
# usage (GPU box, repo root; variants built by tools/build_variant_tb.sh): tools/sweep_config5_r04.sh tb64 tb96 tb128 ...
# Round-4 sweep of BASELINE config 5 (2,000 atoms, r_cut 6 A, three-body cutoff 4 / 6 A) at HEAD: every variant (rows per three-body
# workgroup = waves per workgroup of the moment kernels x 64, LDS window cap) with the three-body MOMENT kernels (default where the
# topology certifies complete partner lists) and with the LIST kernels (option threebody_moments=0), in the arithmetic mode given by
# PREC (default fp32).  -> gpurun_out/config5_sweep_r04.txt
cd "$(dirname "$0")/.."
L=torch-m3gnet_amd/lib
cp $L/libm3gnet_hip.so /tmp/base.so
out=gpurun_out/config5_sweep_r04.jsonl
: > $out
for r in "$@"; do
  cp $L/variants/$r.so $L/libm3gnet_hip.so
  for opt in "" "threebody_moments=0"; do
    M3G_PRECISION=${PREC:-fp32} M3G_ENGINE_OPTIONS=$opt timeout -k 10 200 python tools/time_config5.py "$r ${opt:+lists}" >> $out 2> gpurun_out/config5_sweep_$r.err || echo "rows $r FAILED"
  done
done
cp /tmp/base.so $L/libm3gnet_hip.so
python - <<PY
import json
print("# variant (rows per workgroup[, window cap]) x three-body kernels (moments | lists), precision ${PREC:-fp32}")
print("# variant            kernels  r3   T/atom  step ms   three-body fwd us / rev us per launch")
for line in open("$out"):
    d = json.loads(line)
    f, r = d["threebody_fwd"], d["threebody_rev"]
    lab = d["label"].split()
    print(f'{lab[0]:18s}  {"lists  " if len(lab) > 1 else "moments"}  {d["threebody_cutoff"]:.1f}  {d["triplets_per_atom"]:6.0f}  {d["ms_per_step"]:.3f}    {f["ms_per_launch"]*1e3:6.1f} / {r["ms_per_launch"]*1e3:6.1f}')
PY
