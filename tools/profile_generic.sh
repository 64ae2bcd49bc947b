cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pg -o run -- python3 $GRAFT_REPO_ROOT/tools/time_generic.py > $GRAFT_REPO_ROOT/gpurun_out/r3_generic_prof.txt 2>&1
python3 - <<'PY'
import csv, glob, os
f=glob.glob('/tmp/pg/**/*kernel_stats.csv', recursive=True)[0]
rows=list(csv.DictReader(open(f)))
out=open(os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/r3_generic_kernels.txt','w')
for r in rows[:30]:
    out.write(f"{r['Name'][:80]:80s} calls {r['Calls']:>6s} avg {float(r['AverageNs'])/1e3:9.2f} us total {float(r['TotalDurationNs'])/1e6:9.2f} ms {r['Percentage']}%\n")
PY
