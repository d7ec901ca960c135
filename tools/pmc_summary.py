"""Per-kernel counter totals of tools/pmc_run.sh passes -> a JSON summary under profiles/.

  python tools/pmc_summary.py gpurun_out/<name> profiles/<file>.json --note "..." [--kernels substr,substr]

HBM bytes follow MI355X_MICROARCH.md (HBM section): FETCH_SIZE and WRITE_SIZE are reported in KiB; on gfx950 FETCH_SIZE
tallies a wide coalesced read at half its bytes -> doubled; WRITE_SIZE as reported.  SQ_VALU_MFMA_BUSY_CYCLES /
GRBM_GUI_ACTIVE is reported as given (GUI_ACTIVE is summed over the 8 XCDs by rocprofv3, the busy cycles over all SIMDs:
the JSON keeps both raw sums and the per-SIMD busy fraction = busy / 1024 SIMDs / (GUI_ACTIVE / 8)).
The summary is stamped with the hash of the kernel sources it was measured on (cet_pick_amd.build.source_sha16)."""
import argparse, csv, glob, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cet_pick_amd.build import source_sha16

ap = argparse.ArgumentParser()
ap.add_argument("dir"); ap.add_argument("out")
ap.add_argument("--kernels", default="")
ap.add_argument("--note", default="")
ap.add_argument("--min-calls", type=int, default=1)
ap.add_argument("--per-step", type=float, default=0, help="steps (calls) the profiled run made: adds launches_per_step and hbm_bytes_per_step")
ap.add_argument("--sources", default="", help="file-name prefixes of the kernel sources the stamp covers, e.g. infer_,common")
a = ap.parse_args()
want = [k for k in a.kernels.split(",") if k]
acc = {}
for f in glob.glob(os.path.join(a.dir, "*", "out_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
        name = name.split("(")[0]
        if want and not any(w in name for w in want):
            continue
        d = acc.setdefault(name, {})
        c = d.setdefault(r["Counter_Name"], [0, 0.0])
        c[0] += 1; c[1] += float(r["Counter_Value"])
        d.setdefault("_ns", [0, 0.0])
        if r["Counter_Name"] in ("FETCH_SIZE", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES"):
            d["_ns"][0] += 1; d["_ns"][1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
src = [x for x in a.sources.split(",") if x] or None
out = {"source_prefixes": src, "source_sha16": source_sha16(src), "note": a.note,
       "method": "rocprofv3 --pmc, one pass per counter group with --kernel-trace (tools/pmc_run.sh); FETCH_SIZE / WRITE_SIZE in "
                 "KiB, FETCH_SIZE doubled on gfx950 (MI355X_MICROARCH.md, HBM section)", "kernels": {}}
for name, d in sorted(acc.items()):
    k = {}
    calls = max(v[0] for kk, v in d.items() if kk != "_ns")
    if calls < a.min_calls:
        continue
    k["launches_profiled"] = calls
    if "FETCH_SIZE" in d:
        k["fetch_bytes_per_launch"] = 2 * 1024 * d["FETCH_SIZE"][1] / d["FETCH_SIZE"][0]
    if "WRITE_SIZE" in d:
        k["write_bytes_per_launch"] = 1024 * d["WRITE_SIZE"][1] / d["WRITE_SIZE"][0]
    if "FETCH_SIZE" in d and "WRITE_SIZE" in d:
        k["hbm_bytes_per_launch"] = k["fetch_bytes_per_launch"] + k["write_bytes_per_launch"]
    if "SQ_VALU_MFMA_BUSY_CYCLES" in d and "GRBM_GUI_ACTIVE" in d:
        busy = d["SQ_VALU_MFMA_BUSY_CYCLES"][1] / d["SQ_VALU_MFMA_BUSY_CYCLES"][0]
        gui = d["GRBM_GUI_ACTIVE"][1] / d["GRBM_GUI_ACTIVE"][0]
        k["mfma_busy_cycles_per_launch"] = busy
        k["grbm_gui_active_per_launch"] = gui
        k["mfma_busy_frac_per_simd"] = busy / 1024.0 / (gui / 8.0) if gui else None
    if d["_ns"][0]:
        k["avg_us_while_profiled"] = d["_ns"][1] / d["_ns"][0] / 1e3
    if a.per_step:
        k["launches_per_step"] = calls / a.per_step
    out["kernels"][name] = k
if a.per_step:
    out["steps_profiled"] = a.per_step
    out["hbm_bytes_per_step"] = sum(k["hbm_bytes_per_launch"] * k["launches_per_step"] for k in out["kernels"].values()
                                    if "hbm_bytes_per_launch" in k)
json.dump(out, open(a.out, "w"), indent=1)
print(json.dumps(out, indent=1)[:6000])
