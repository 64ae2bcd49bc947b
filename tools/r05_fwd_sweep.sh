cd $GRAFT_REPO_ROOT
for st in 1024 2048 4096 8192 32768; do
  echo "small_tiles_fwd=$st" >> gpurun_out/r05_small_fwd_sweep.txt
  M3G_SMALL_TILES_FWD=$st python3 tools/time_small_systems.py fp32 5 6 7 8 10 12 2>/dev/null >> gpurun_out/r05_small_fwd_sweep.txt
done
cat gpurun_out/r05_small_fwd_sweep.txt
