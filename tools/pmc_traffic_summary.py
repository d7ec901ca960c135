"""profiles/r01_conv_traffic.json from the two PMC passes of tools/pmc_traffic.sh (gpurun_out/pmc_traffic/{FETCH_SIZE,WRITE_SIZE})."""
import csv, glob, json, os, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MAIN = ("conv_igemm_kernel", "stem_fwd_kernel", "stem_fwd_bf3_kernel", "stem_wgrad_kernel")
AUX = ("splitk_reduce_kernel", "stem_wgrad_reduce_kernel", "stem_wprep_kernel")


def total(counter):
    f = glob.glob(os.path.join(ROOT, "gpurun_out", "pmc_traffic", counter, "**", "*counter_collection.csv"), recursive=True)
    assert f, "no counter_collection.csv for " + counter
    tot, calls, disp = 0.0, 0, 0
    for r in csv.DictReader(open(f[0])):
        name = r["Kernel_Name"]
        if r["Counter_Name"] != counter or not any(k in name for k in MAIN + AUX):
            continue
        tot += float(r["Counter_Value"])
        disp += 1
        calls += any(k in name for k in MAIN)
    return tot, calls, disp


fetch_kb, calls, disp = total("FETCH_SIZE")
write_kb, calls_w, _ = total("WRITE_SIZE")
assert calls == calls_w, (calls, calls_w)
hbm = 2 * fetch_kb * 1024 + write_kb * 1024
out = {
    "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, tools/pmc_traffic.sh) over `bench.py --no-secondary "
              "--no-cpu-baseline --steps 6 --warmup 2 --no-graph`; kernels conv_igemm / stem_* / splitk_reduce; a conv call = one "
              "dispatch of the main kernel (its reduce / weight-preparation dispatches are counted into it)",
    "fetch_size_kb_raw": fetch_kb, "write_size_kb": write_kb, "dispatches": disp, "conv_calls": calls,
    "gfx950_correction": "FETCH_SIZE doubled (MI355X_MICROARCH.md: wide coalesced reads are tallied at half their bytes); "
                         "WRITE_SIZE as reported",
    "hbm_bytes_per_conv_call": hbm / calls, "conv_calls_per_step": 75, "hbm_bytes_per_step": hbm / calls * 75,
}
json.dump(out, open(os.path.join(ROOT, "profiles", "r01_conv_traffic.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
