cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for n in 2 6; do
  rocprofv3 --kernel-trace --output-format csv -d /tmp/tr$n -- python3 $R/tools/small_step_trace.py fp32 $n > $R/gpurun_out/r05_trace_n$n.log 2>&1
  python3 $R/tools/step_sequence.py /tmp/tr$n 2 > $R/gpurun_out/r05_seq_n$n.txt
done
cut -c1-120 $R/gpurun_out/r05_seq_n2.txt
