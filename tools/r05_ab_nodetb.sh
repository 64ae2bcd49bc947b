cd $GRAFT_REPO_ROOT
for v in 1 0 1 0; do
  python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-secondary --engine-option fuse_node_tb=$v 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('fuse_node_tb=$v', round(d['ms_per_step'],4), round(d['ms_per_step_min'],4), d['clock_mhz'])"
done
