# round 5, final build: the GPU test suite, the default bench line, the step sequences and the MD figures
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
rm -f gpurun_out/parity_margins.txt gpurun_out/stress_case_margins.txt gpurun_out/small_vs_large_margins.txt gpurun_out/config3_vs_reference.txt
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/r05_gputest_final.log 2>&1; rc=$?
tail -3 gpurun_out/r05_gputest_final.log
[ $rc -eq 0 ] || exit $rc
python bench.py > gpurun_out/r05_bench_default_run.json 2> gpurun_out/r05_bench_default_run.err || exit 1
python -c "
import json; d=json.load(open('gpurun_out/r05_bench_default_run.json')); print(d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('traffic'), d['cpu_baseline']['value']); print(d['config']['secondary'])"
# the reference's own Legendre backward (three-body list kernels): what the option costs at the headline size
python bench.py --engine-option legendre_backward=1 --no-secondary --no-cpu-baseline > gpurun_out/r05_bench_legendre_backward_reference.json 2>/dev/null
python -c "
import json; d=json.load(open('gpurun_out/r05_bench_legendre_backward_reference.json')); print('legendre_backward=1:', d['ms_per_step'], 'ms/step')"
cd /tmp && export TMPDIR=/tmp
for n in 2 6; do
  rocprofv3 --kernel-trace --output-format csv -d /tmp/tr$n -- python3 $R/tools/small_step_trace.py fp32 $n > /dev/null 2>&1
  python3 $R/tools/step_sequence.py /tmp/tr$n 2 > $R/gpurun_out/r05_seq_n$n.txt
done
rocprofv3 --kernel-trace --output-format csv -d /tmp/seq -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary > /dev/null 2>&1
python3 $R/tools/step_sequence.py /tmp/seq 4 > $R/gpurun_out/r05_step_sequence_fp32.txt
cd $R
python3 tools/time_small_systems.py fp32 2 3 4 5 6 8 10 > gpurun_out/r05_small_cells_timing.txt 2>/dev/null
for m in reuse refill rebuild; do python tools/profile_md_iteration.py $m 30 2>/dev/null | tail -1; done > gpurun_out/r05_md_times.txt
cat gpurun_out/r05_small_cells_timing.txt gpurun_out/r05_md_times.txt
tail -3 gpurun_out/r05_step_sequence_fp32.txt
