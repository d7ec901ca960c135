mkdir -p gpurun_out/r03t
for i in 1 2; do
for v in 2 1; do
MI_STEM_WGRAD_OCC=$v timeout -k 10 200 python bench.py --no-conv-profile --no-secondary --no-cpu-baseline --steps 300 --warmup 20 > gpurun_out/r03t/occ$v.$i.json 2> gpurun_out/r03t/occ$v.$i.err || exit 1
python - <<PY
import json
d=json.loads(open("gpurun_out/r03t/occ$v.$i.json").read().strip().splitlines()[-1])
print("stem_wgrad_occ=$v run $i ms_per_step", d["ms_per_step"])
PY
done; done
