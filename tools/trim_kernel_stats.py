"""Shorten a rocprofv3 --stats kernel_stats.csv for profiles/: argument lists dropped, rocprim internals collapsed."""
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
print(f"# {sys.argv[2]}")
print("Name,Calls,TotalDurationNs,AverageNs,Percentage")
for r in rows:
    n = r["Name"].split("(")[0]
    n = "rocprim::radix_sort<...>" if "rocprim" in n else n
    n = re.sub(r"^void ", "", n)
    print(f"\"{n}\",{r['Calls']},{r['TotalDurationNs']},{float(r['AverageNs']):.1f},{r['Percentage']}")
