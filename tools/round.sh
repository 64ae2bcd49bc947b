#!/bin/bash
# The round's routine GPU-box command sets behind one entry point (replaces the 24 one-off r05_*.sh wrappers).
#   usage (on the GPU box, from the repo root):  bash tools/round.sh <verb> [round tag, default r06] [args]
#     ab [tag] [bench args]   quick A/B figures of the current build: the headline step (100 steps), its stage times, small / mid cells
#     small [tag] [n ...]     step latency of n x n x n fcc Cu cells (default 2 3 4 5 6 7 8 10) + config 5
#     seq [tag] [n ...]       one step's kernel sequence (rocprofv3 --kernel-trace) of n x n x n cells and of the headline cell
#     collect [tag]           the three arithmetic modes' profile sets (kernel stats, PMC traffic, SQ counters) at the CURRENT kernel
#                             sources -- once per round, at the final sources -- copied to profiles/<tag>_*
#     final [tag]             GPU test suite, default bench line, step sequences, small-cell and MD figures -> gpurun_out/<tag>_*
#     soak [tag]              determinism repeats + graph-builder fuzz on the final tree
# Intermediate A/Bs write to gpurun_out/ only; what should be judged is copied into profiles/ by `collect` / by hand.
set -o pipefail
verb=${1:?verb}; tag=${2:-r06}; shift; shift 2>/dev/null
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
case $verb in
ab)
  python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-secondary "$@" 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('ms/step', round(d['ms_per_step'],4), 'min', round(d['ms_per_step_min'],4), 'MHz', d['clock_mhz'], 'launches', d['kernel_launches_per_step']); print(d['config']['stage_ms_per_step'])"
  python tools/time_small_systems.py fp32 2 5 6 8 10 2>/dev/null
  python tools/time_config5.py 2>/dev/null | tail -2
  ;;
small)
  python tools/time_small_systems.py fp32 ${@:-2 3 4 5 6 7 8 10} 2>/dev/null | tee gpurun_out/${tag}_small_cells_timing.txt
  python tools/time_config5.py 2>/dev/null | tail -2 | tee -a gpurun_out/${tag}_small_cells_timing.txt
  ;;
seq)
  cd /tmp && export TMPDIR=/tmp
  for n in ${@:-2 6}; do
    rm -rf /tmp/tr$n
    rocprofv3 --kernel-trace --output-format csv -d /tmp/tr$n -- python3 $R/tools/small_step_trace.py fp32 $n > /dev/null 2>&1
    python3 $R/tools/step_sequence.py /tmp/tr$n 2 > $R/gpurun_out/${tag}_step_sequence_$((4*n*n*n))_atoms.txt
    tail -2 $R/gpurun_out/${tag}_step_sequence_$((4*n*n*n))_atoms.txt
  done
  rm -rf /tmp/seq
  rocprofv3 --kernel-trace --output-format csv -d /tmp/seq -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary > /dev/null 2>&1
  python3 $R/tools/step_sequence.py /tmp/seq 4 > $R/gpurun_out/${tag}_step_sequence_fp32.txt
  tail -3 $R/gpurun_out/${tag}_step_sequence_fp32.txt
  ;;
collect)
  rm -f gpurun_out/pmc_hbm_traffic.json gpurun_out/pmc_sq_counters.json
  for mode in fp32 f16x3 bf16x3; do
    bash tools/collect_profiles.sh ${tag}_$mode --precision $mode --no-secondary --no-cpu-baseline || exit 1
    bash tools/collect_sq_counters.sh ${tag}_$mode --precision $mode --no-secondary || exit 1
    cp gpurun_out/${tag}_$mode/kernel_stats.csv profiles/${tag}_${mode}_kernel_stats.csv
    echo "[$tag] $mode set done"
  done
  cp gpurun_out/pmc_hbm_traffic.json profiles/${tag}_pmc_hbm_traffic.json
  cp gpurun_out/pmc_sq_counters.json profiles/${tag}_pmc_sq_counters.json
  ;;
final)
  rm -f gpurun_out/parity_margins.txt gpurun_out/stress_case_margins.txt gpurun_out/small_vs_large_margins.txt gpurun_out/config3_vs_reference.txt gpurun_out/mid_size_margins.txt
  timeout -k 10 1100 python -m pytest tests -m gpu -x -q > gpurun_out/${tag}_gputest_final.log 2>&1; rc=$?
  tail -3 gpurun_out/${tag}_gputest_final.log
  [ $rc -eq 0 ] || exit $rc
  python bench.py > gpurun_out/${tag}_bench_default_run.json 2> gpurun_out/${tag}_bench_default_run.err || exit 1
  python -c "
import json; d=json.load(open('gpurun_out/${tag}_bench_default_run.json')); print(d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('traffic'), d['cpu_baseline']['value'], d['cpu_baseline']['cores']); print({k: v for k, v in d['config'].items() if k.startswith('sec_')})"
  bash tools/round.sh seq $tag 2 6
  python3 tools/time_small_systems.py fp32 2 3 4 5 6 8 10 > gpurun_out/${tag}_small_cells_timing.txt 2>/dev/null
  for m in reuse refill rebuild; do python tools/profile_md_iteration.py $m 30 2>/dev/null | tail -1; done > gpurun_out/${tag}_md_times.txt
  cat gpurun_out/${tag}_small_cells_timing.txt gpurun_out/${tag}_md_times.txt
  ;;
soak)
  python tools/stress_determinism.py > gpurun_out/${tag}_stress_determinism.txt 2>&1; cat gpurun_out/${tag}_stress_determinism.txt
  timeout -k 10 600 python tests/checkers/fuzz_graph_build.py 300 > gpurun_out/${tag}_fuzz_graph_build.txt 2>&1; tail -1 gpurun_out/${tag}_fuzz_graph_build.txt
  ;;
*) echo "unknown verb $verb"; exit 2 ;;
esac
