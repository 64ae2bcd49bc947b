#!/bin/bash
# usage (on the GPU box, from the repo root): tools/collect_profiles.sh <tag>
# Writes gpurun_out/<tag>/{bench.json, bench_under_rocprof.json, kernel_stats.csv, pmc_hbm_traffic.json, pmc_hbm_traffic.txt}:
# the plain bench line, the same command under `rocprofv3 --kernel-trace --stats`, and HBM traffic per launch from two
# separate --pmc passes (FETCH_SIZE / WRITE_SIZE cannot share one on gfx950).  Copy what should be judged into profiles/.
set -e -o pipefail
tag=$1
shift
extra="$@"   # extra bench.py arguments, e.g. --precision bf16x3 --no-secondary
root=$(pwd)
out=$root/gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
python3 bench.py $extra > $out/bench.json 2> $out/bench.err
echo "[collect] bench done"
cd /tmp
rm -rf /tmp/prof_$tag && mkdir -p /tmp/prof_$tag
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag/stats -o run -- python3 $root/bench.py --steps 10 --warmup 3 --no-cpu-baseline $extra > $out/bench_under_rocprof.json 2> $out/rocprof_stats.err
echo "[collect] stats pass done"
python3 $root/tools/trim_kernel_stats.py $(find /tmp/prof_$tag/stats -name '*kernel_stats.csv' | head -1) "rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline $extra ($tag build)" > $out/kernel_stats.csv
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/prof_$tag/fetch -o run -- python3 $root/bench.py --steps 3 --warmup 1 --no-cpu-baseline $extra > /dev/null 2> $out/rocprof_fetch.err
echo "[collect] fetch pass done"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/prof_$tag/write -o run -- python3 $root/bench.py --steps 3 --warmup 1 --no-cpu-baseline $extra > /dev/null 2> $out/rocprof_write.err
echo "[collect] write pass done"
{
  echo "# rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, --kernel-trace), bench.py --steps 3 --warmup 1 --no-cpu-baseline, $tag build"
  mode=$(echo "$extra" | sed -n 's/.*--precision \([a-z0-9]*\).*/\1/p'); mode=${mode:-fp32}   # bench.py's default mode
  # merged into one stamped file per round (both modes), which bench.py reads: $root/gpurun_out/pmc_hbm_traffic.json
  python3 $root/tools/pmc_traffic.py $(find /tmp/prof_$tag/fetch -name '*counter_collection.csv' | head -1) $(find /tmp/prof_$tag/write -name '*counter_collection.csv' | head -1) $root/gpurun_out/pmc_hbm_traffic.json $mode
} > $out/pmc_hbm_traffic.txt
echo "[collect] done"
