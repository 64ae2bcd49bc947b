cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_small_path.py tests/test_gpu_determinism.py tests/test_gpu_properties.py tests/test_gpu_parity.py -x -q 2>&1 | tail -3
for v in 1 0 1 0 1 0; do
  python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-secondary --engine-option dp1_by_dst=$v 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('dp1_by_dst=$v', round(d['ms_per_step'],4), round(d['ms_per_step_min'],4), d['clock_mhz'], d['config']['stage_ms_per_step']['node_rev'], d['config']['stage_ms_per_step']['edge_rev_fused'])"
done
