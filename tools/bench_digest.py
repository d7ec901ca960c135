"""Print the interesting fields of a bench.py JSON line."""
import json, sys
d = json.load(open(sys.argv[1]))
print("value %.1f %s  ms/step %.4f  n_gpus %d steps %d  rccl_ranks %s" % (d["value"], d["unit"], d["ms_per_step"], d["n_gpus"], d["steps"], d.get("rccl_ranks")))
r = d["roofline"]
print("roofline: achieved %.1f %s  peak %.1f  frac %.4f  kernel_ms/step %.4f  launches/step %s  traffic %s" % (
    r["achieved"], r["unit"], r["peak"], r["frac"], r.get("kernel_ms_per_step", 0), r.get("launches_per_step"), r.get("traffic")))
for k in ("step_graph_nodes", "f32_mfma_step", "cpu_baseline"):
    if k in d:
        print(k, d[k])
s = d.get("secondary") or {}
for k in ("decode_sigmoid_nms_topk", "dog_pick"):
    if k in s:
        e = s[k]
        print("%s: ms %.4f (eager %.4f, graph %.4f)  frac %.4f  traffic %s  [%s]" % (k, e["ms"], e["ms_eager"], e["ms_hipgraph"], e["roofline"]["frac"], e["roofline"]["traffic"], e["roofline"]["traffic_note"]))
        for kk in e:
            if kk.startswith("cpu"):
                print("   ", kk, "%.3g %s, %s" % (e[kk]["value"], e[kk]["unit"], e[kk]["sample"][:70]))
if "detector" in s:
    for k, v in s["detector"].items():
        print("detector.%s: %s" % (k, {a: (round(b, 3) if isinstance(b, float) else b) for a, b in v.items() if not isinstance(b, (list, dict, str))}))
