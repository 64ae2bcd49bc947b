cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_md.py -x -q 2>&1 | tail -2
python tools/time_md_small.py 2 3 4 2>/dev/null | grep "replay=0"
