# round 5: the three profile sets at the current kernel sources, then (with those sets in place) the test suite, the default bench line
# and the step / MD figures
cd $GRAFT_REPO_ROOT
bash tools/r05_collect.sh > gpurun_out/r05_collect.log 2>&1 || { tail -5 gpurun_out/r05_collect.log; exit 1; }
cp gpurun_out/pmc_hbm_traffic.json profiles/r05_pmc_hbm_traffic.json
cp gpurun_out/pmc_sq_counters.json profiles/r05_pmc_sq_counters.json
bash tools/r05_final.sh
