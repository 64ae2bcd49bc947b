cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_small_path.py tests/test_gpu_determinism.py tests/test_gpu_properties.py tests/test_gpu_parity.py tests/test_gpu_md.py tests/test_gpu_c_abi.py -x -q 2>&1 | tail -3
python3 tools/time_small_systems.py fp32 2 6 10 2>/dev/null
for v in 1 0 1 0; do
  python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-secondary --engine-option small_launches=$v 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('small_launches=$v', round(d['ms_per_step'],4), round(d['ms_per_step_min'],4), d['clock_mhz'])"
done
