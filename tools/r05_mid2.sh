cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_small_path.py -x -q 2>&1 | tail -2
for st in 1024 4096 32768; do
  echo "small_tiles=$st (forward and reverse)"
  M3G_SMALL_TILES=$st python3 tools/time_small_systems.py fp32 2 4 5 6 7 8 10 2>/dev/null
done > gpurun_out/r05_small_tiles_sweep2.txt
cat gpurun_out/r05_small_tiles_sweep2.txt
