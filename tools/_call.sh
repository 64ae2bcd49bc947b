cd $GRAFT_REPO_ROOT
L=torch-m3gnet_amd/lib
cp $L/libm3gnet_hip.so /tmp/base.so
for v in base g_fastdiv base g_fastdiv; do
  if [ "$v" = base ]; then cp /tmp/base.so $L/libm3gnet_hip.so; else cp $L/variants/$v.so $L/libm3gnet_hip.so; fi
  python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$v', round(d['ms_per_step'],4), {k: d['config']['stage_ms_per_step'][k] for k in ('geometry_basis','node_rev','geometry_rev_forces','readout')})"
done
cp /tmp/base.so $L/libm3gnet_hip.so
