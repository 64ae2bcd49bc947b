cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for tb in 4.0 6.0; do
  rocprofv3 --kernel-trace --output-format csv -d /tmp/c5_$tb -- python3 $R/tools/config5_trace.py $tb > /dev/null 2>&1
  python3 $R/tools/step_sequence.py /tmp/c5_$tb 2 | cut -c1-105 > $R/gpurun_out/r05_config5_sequence_r3_$tb.txt
  cat $R/gpurun_out/r05_config5_sequence_r3_$tb.txt
done
