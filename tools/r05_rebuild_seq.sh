cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --output-format csv -d /tmp/mdb -- python3 $R/tools/profile_md_iteration.py rebuild 6 > /dev/null 2>&1
python3 - <<'PY'
import csv, glob, os
rows = []
for p in glob.glob('/tmp/mdb/**/*kernel_trace.csv', recursive=True):
    rows += list(csv.DictReader(open(p)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# one iteration = from a k_struct_info (start of a search) to the next
idx = [i for i, r in enumerate(rows) if 'k_struct_info' in r['Kernel_Name']]
a, b = idx[-2], idx[-1]
out = open(os.environ['GRAFT_REPO_ROOT'] + '/gpurun_out/r05_md_rebuild_sequence.txt', 'w')
prev = None
tot = 0.0
for r in rows[a:b]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    gap = (s - prev) / 1e3 if prev else 0.0
    out.write(f"{(e - s) / 1e3:9.1f} us  gap {gap:7.1f}  {r['Kernel_Name'][:100]}\n")
    tot += (e - s) / 1e3
    prev = e
span = (int(rows[b]['Start_Timestamp']) - int(rows[a]['Start_Timestamp'])) / 1e3
out.write(f"iteration span {span:.1f} us, sum of kernel durations {tot:.1f} us, {b - a} dispatches\n")
PY
awk '/k_geometry</{exit} {print}' $R/gpurun_out/r05_md_rebuild_sequence.txt | cut -c1-120
tail -1 $R/gpurun_out/r05_md_rebuild_sequence.txt
