"""Per-kernel averages of SQ counters from tools/pmc_run.sh passes (one row per kernel, one column per counter, per launch)."""
import csv, glob, os, sys
d = sys.argv[1]
want = sys.argv[2].split(",") if len(sys.argv) > 2 else []
acc, cols = {}, []
for f in sorted(glob.glob(os.path.join(d, "*", "out_counter_collection.csv"))):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        if want and not any(w in name for w in want):
            continue
        c = r["Counter_Name"]
        if c not in cols:
            cols.append(c)
        e = acc.setdefault(name, {}).setdefault(c, [0, 0.0])
        e[0] += 1; e[1] += float(r["Counter_Value"])
print("%-52s" % "kernel" + "".join("%16s" % c.replace("SQ_", "")[:15] for c in cols))
for k in sorted(acc):
    print("%-52s" % k[:52] + "".join("%16.4g" % (acc[k][c][1] / acc[k][c][0]) if c in acc[k] else "%16s" % "-" for c in cols))
