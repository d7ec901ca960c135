"""Print a rocprofv3 kernel_stats.csv compactly: calls, average us, total us."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    name = r["Name"]
    for junk in ("(anonymous namespace)::", "void "):
        name = name.replace(junk, "")
    print("%-60s %5s  avg %9.2f us  tot %10.1f us  %5s%%" % (name[:60], r["Calls"], float(r["AverageNs"]) / 1e3,
                                                             float(r["TotalDurationNs"]) / 1e3, r["Percentage"][:5]))
