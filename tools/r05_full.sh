cd $GRAFT_REPO_ROOT
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/r05_gputest.log 2>&1; rc=$?
tail -5 gpurun_out/r05_gputest.log
[ $rc -eq 0 ] || exit $rc
bash tools/r05_small_check.sh
python bench.py --steps 50 --warmup 10 --no-cpu-baseline > gpurun_out/r05_bench_quick.json 2> gpurun_out/r05_bench_quick.err
python -c "
import json; d=json.load(open('gpurun_out/r05_bench_quick.json')); print(d['ms_per_step'], d.get('roofline',{}).get('frac'))"
