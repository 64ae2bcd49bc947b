# round 5: small-path tests, then the step sequence of the 32- and 864-atom cells and the small-cell timings
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_small_path.py tests/test_gpu_determinism.py -x -q > gpurun_out/r05_small_tests.log 2>&1 || { tail -30 gpurun_out/r05_small_tests.log; exit 1; }
tail -3 gpurun_out/r05_small_tests.log
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for n in 2 6; do
  rocprofv3 --kernel-trace --output-format csv -d /tmp/small$n -- python3 $R/tools/small_step_trace.py fp32 $n > /dev/null 2>&1
  python3 $R/tools/step_sequence.py /tmp/small$n 2 > $R/gpurun_out/r05_seq_n$n.txt
done
python3 $R/tools/time_small_systems.py fp32 2 3 4 6 8 > $R/gpurun_out/r05_small.txt 2>&1
cat $R/gpurun_out/r05_small.txt
