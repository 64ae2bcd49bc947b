# round 5: the three arithmetic modes' profile sets at the final kernel sources (collected ONCE), then the default bench line
cd $GRAFT_REPO_ROOT
rm -f gpurun_out/pmc_hbm_traffic.json gpurun_out/pmc_sq_counters.json
for mode in fp32 f16x3 bf16x3; do
  bash tools/collect_profiles.sh r05_$mode --precision $mode --no-secondary --no-cpu-baseline || exit 1
  bash tools/collect_sq_counters.sh r05_$mode --precision $mode --no-secondary || exit 1
  echo "[r05] $mode set done"
done
