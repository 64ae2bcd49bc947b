cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_md.py tests/test_gpu_graph_build.py -x -q > gpurun_out/r05_md_tests.log 2>&1 || { tail -40 gpurun_out/r05_md_tests.log; exit 1; }
tail -3 gpurun_out/r05_md_tests.log
for m in reuse refill rebuild; do python tools/profile_md_iteration.py $m 30 2>/dev/null | tail -1; done > gpurun_out/r05_md_times.txt
cat gpurun_out/r05_md_times.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d /tmp/mdr -- python3 $GRAFT_REPO_ROOT/tools/profile_md_iteration.py refill 6 > /dev/null 2>&1
python3 $GRAFT_REPO_ROOT/tools/step_sequence.py /tmp/mdr 2 k_verlet_prep > $GRAFT_REPO_ROOT/gpurun_out/r05_md_refill_sequence.txt
