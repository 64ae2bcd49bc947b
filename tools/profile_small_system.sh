cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/sp -o run -- python3 $GRAFT_REPO_ROOT/tools/time_small_systems.py f16x3 2 > $GRAFT_REPO_ROOT/gpurun_out/r3_small_prof.txt 2>&1
python3 - <<'PY'
import csv, glob, os
f=glob.glob('/tmp/sp/**/*kernel_stats.csv', recursive=True)[0]
rows=list(csv.DictReader(open(f)))
out=open(os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/r3_small_kernels.txt','w')
for r in rows[:40]:
    out.write(f"{r['Name'][:70]:70s} calls {r['Calls']:>6s} avg {float(r['AverageNs'])/1e3:8.2f} us  {r['Percentage']}%\n")
PY
