"""Timeline of ONE graph-replayed training step from a rocprofv3 --kernel-trace CSV: per kernel start (us from the step's
first kernel), duration, queue, and the gap to the previous kernel on the same queue; then totals per kernel name and the
busy / idle time of the union of all queues.   python tools/trace_timeline.py <out_kernel_trace.csv> [step index from the end]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
back = int(sys.argv[2]) if len(sys.argv) > 2 else 2
for r in rows:
    r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
rows.sort(key=lambda r: r["s"])
# a step starts at every ema_kernel launch (first kernel of the captured step is the momentum update on the side stream
# or the stem on the main stream): split at stem_wprep / ema markers
starts = [i for i, r in enumerate(rows) if "sgd_kernel" in r["Kernel_Name"]]
# step k = kernels after sgd+prep of step k-1 up to and including sgd of step k
a, b = starts[-back - 1], starts[-back]
# include the image re-cut launch that follows the sgd kernel
seg = rows[a + 1:b + 2]
t0 = seg[0]["s"]
last_end = {}
print("%9s %8s %6s %4s  %s" % ("start_us", "dur_us", "gap", "q", "kernel"))
for r in seg:
    q = r["Queue_Id"]
    gap = (r["s"] - last_end[q]) / 1e3 if q in last_end else 0.0
    last_end[q] = r["e"]
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0][:60]
    print("%9.1f %8.1f %6.1f %4s  %s  grid %s" % ((r["s"] - t0) / 1e3, (r["e"] - r["s"]) / 1e3, gap, q, name, r["Grid_Size_X"]))
print("step span %.1f us, %d kernels" % ((max(r["e"] for r in seg) - t0) / 1e3, len(seg)))
# union busy time
ev = sorted([(r["s"], 1) for r in seg] + [(r["e"], -1) for r in seg])
busy = 0; depth = 0; prev = None
for t, d in ev:
    if depth > 0: busy += t - prev
    depth += d; prev = t
print("union busy %.1f us" % (busy / 1e3))
tot = collections.defaultdict(lambda: [0, 0.0])
for r in seg:
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0][:60]
    tot[name][0] += 1; tot[name][1] += (r["e"] - r["s"]) / 1e3
for k, v in sorted(tot.items(), key=lambda kv: -kv[1][1]):
    print("%8.1f us %3d  %s" % (v[1], v[0], k))
