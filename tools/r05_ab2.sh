cd $GRAFT_REPO_ROOT
for v in 128 4096 128 4096; do
  python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-secondary --engine-option split_node_tiles=$v 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('split_node_tiles=$v', round(d['ms_per_step'],4), round(d['ms_per_step_min'],4), d['clock_mhz'], {k:v for k,v in d['config']['stage_ms_per_step'].items() if k in ('node_pre','readout','geometry_basis')})"
done
