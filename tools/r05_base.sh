cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for n in 2 6; do
  rocprofv3 --kernel-trace --output-format csv -d /tmp/small$n -- python3 $R/tools/small_step_trace.py fp32 $n > /dev/null 2>&1
  python3 $R/tools/step_sequence.py /tmp/small$n 2 > $R/gpurun_out/r05_base_seq_n$n.txt
done
python3 $R/tools/time_small_systems.py fp32 2 3 4 6 8 > $R/gpurun_out/r05_base_small.txt 2>&1
python3 $R/tools/time_md_small.py > $R/gpurun_out/r05_base_md_small.txt 2>&1
cat $R/gpurun_out/r05_base_small.txt
