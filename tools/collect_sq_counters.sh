#!/bin/bash
# usage (GPU box, repo root): tools/collect_sq_counters.sh <tag>  -> gpurun_out/<tag>/pmc_sq_counters.txt
# Two rocprofv3 --pmc passes (kernel trace only) of a short bench run; mean counter value per launch for the m3g kernels.
set -e -o pipefail
tag=$1
shift
extra="$@"   # extra bench.py arguments, e.g. --precision bf16x3 --no-secondary
root=$(pwd)
out=$root/gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
cd /tmp
rm -rf /tmp/sq_$tag
P1="SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CU_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAVE_CYCLES"
P2="SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAVES"
rocprofv3 --kernel-trace --pmc $P1 --output-format csv -d /tmp/sq_$tag/p1 -o run -- python3 $root/bench.py --steps 3 --warmup 1 --no-cpu-baseline $extra > /dev/null 2> $out/rocprof_sq1.err
echo "[sq] pass 1 done"
rocprofv3 --kernel-trace --pmc $P2 --output-format csv -d /tmp/sq_$tag/p2 -o run -- python3 $root/bench.py --steps 3 --warmup 1 --no-cpu-baseline $extra > /dev/null 2> $out/rocprof_sq2.err
echo "[sq] pass 2 done"
{
  echo "# rocprofv3 --kernel-trace --pmc <SQ counters, two passes> -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline $extra ($tag build)"
  echo "# mean counter value per launch (tools/pmc_report.py); SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_* are in units of 4 cycles"
  python3 $root/tools/pmc_report.py $(find /tmp/sq_$tag/p1 -name '*counter_collection.csv' | head -1) k_ | grep -v "k_convert\|k_low\|k_lower\|k_active\|k_tb_win\|k_pair\|k_compact\|k_partner"
  python3 $root/tools/pmc_report.py $(find /tmp/sq_$tag/p2 -name '*counter_collection.csv' | head -1) k_ | grep -v "k_convert\|k_low\|k_lower\|k_active\|k_tb_win\|k_pair\|k_compact\|k_partner"
} > $out/pmc_sq_counters.txt
mode=$(echo "$extra" | sed -n 's/.*--precision \([a-z0-9]*\).*/\1/p'); mode=${mode:-fp32}   # bench.py's default mode
# merged into one stamped file per round (all modes), which bench.py reads: $root/gpurun_out/pmc_sq_counters.json
python3 $root/tools/pmc_sq_json.py $(find /tmp/sq_$tag/p1 -name '*counter_collection.csv' | head -1) $(find /tmp/sq_$tag/p2 -name '*counter_collection.csv' | head -1) $root/gpurun_out/pmc_sq_counters.json $mode | tee -a $out/pmc_sq_counters.txt
echo "[sq] done"
