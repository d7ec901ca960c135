"""Average kernel durations of a rocprofv3 kernel trace in chunks of N consecutive calls per kernel name
(a probe script calls each configuration N times in a row).  usage: trace_chunks.py trace.csv N [name-filter]"""
import csv, sys
from collections import OrderedDict
rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2])
flt = sys.argv[3] if len(sys.argv) > 3 else ""
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
by = OrderedDict()
for r in rows:
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
    if flt and flt not in name:
        continue
    by.setdefault(name[:50], []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in by.items():
    ch = [v[i:i + n] for i in range(0, len(v), n)]
    print("%-50s %s" % (k, " ".join("%.1f" % (sorted(c)[len(c) // 2]) for c in ch)))
