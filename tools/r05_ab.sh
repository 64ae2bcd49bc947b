cd $GRAFT_REPO_ROOT
python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],4), round(d['ms_per_step_min'],4), d['clock_mhz']); print({k:v for k,v in d['config']['stage_ms_per_step'].items()})"
python tools/time_small_systems.py fp32 2 6 2>/dev/null
