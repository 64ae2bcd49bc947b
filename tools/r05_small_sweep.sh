cd $GRAFT_REPO_ROOT
for st in 0 1024 4096 32768; do
  echo "small_tiles=$st" >> gpurun_out/r05_small_sweep.txt
  M3G_SMALL_TILES=$st python3 tools/time_small_systems.py fp32 3 4 5 6 7 8 10 12 2>/dev/null >> gpurun_out/r05_small_sweep.txt
done
cat gpurun_out/r05_small_sweep.txt
