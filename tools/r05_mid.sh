cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for n in 2 4 6; do
  M3G_SMALL_TILES=32768 rocprofv3 --kernel-trace --output-format csv -d /tmp/mid$n -- python3 $R/tools/small_step_trace.py fp32 $n > /dev/null 2>&1
  python3 $R/tools/step_sequence.py /tmp/mid$n 2 | grep -E "edge|step span" | cut -c1-110 > $R/gpurun_out/r05_mid_split_n$n.txt
  cat $R/gpurun_out/r05_mid_split_n$n.txt
done
