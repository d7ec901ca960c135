#!/bin/bash
# Summaries of tools/refresh_profiles.sh's raw output -> profiles/<tag>_* (stamped with the kernel-source hashes).
set -e
cd "$(dirname "$0")/.."
T=${1:-r06}
O=gpurun_out/$T"_final"
cp $O/train/out_kernel_stats.csv profiles/${T}_train_kernel_stats.csv
cp $O/infer/out_kernel_stats.csv profiles/${T}_infer_kernel_stats.csv
if [ -s $O/c5/out_kernel_stats.csv ]; then cp $O/c5/out_kernel_stats.csv profiles/${T}_c5_kernel_stats.csv; fi
python tools/pmc_summary.py $O/pmc_infer profiles/${T}_infer_traffic.json --sources infer_,common --per-step 5 \
  --note "tools/prof_infer.py both 3 (2 synchronized calls + 3 back to back of each chain): 5 fused decodes (logits 128x256x256, K=900) + 5 DoG picks (256x512x512, sigma 3/5); per-launch averages" > /dev/null
mkdir -p $O/pmc_traffic $O/pmc_busy
rm -rf $O/pmc_traffic/* $O/pmc_busy/*
cp -r $O/pmc_train/FETCH_SIZE $O/pmc_train/WRITE_SIZE $O/pmc_traffic/
cp -r "$O/pmc_train/SQ_VALU_MFMA_BUSY_CYCLES+GRBM_GUI_ACTIVE" $O/pmc_busy/
python tools/pmc_summary.py $O/pmc_traffic profiles/${T}_conv_traffic.json --sources conv_ --per-step 9 --kernels conv_igemm,stem_,direct3,splitk,cube2,pair_wgrad,s2_,small_gemm \
  --note "bench.py --no-secondary --no-cpu-baseline --no-conv-profile --no-graph --steps 6 --warmup 3 (9 eager steps): the conv family of the MoCo-3D step" > /dev/null
python tools/pmc_summary.py $O/pmc_busy profiles/${T}_mfma_busy.json --sources conv_,loss_ --per-step 9 --kernels conv_igemm,stem_,direct3,cube2,pair_wgrad,s2_,small_gemm \
  --note "same run; SQ_VALU_MFMA_BUSY_CYCLES and GRBM_GUI_ACTIVE in one pass" > /dev/null
# round 6: SimSiam-2D step (24 steps per run of tools/bench_simsiam2d.py --only-step), unet_4 forward (5 forwards per run of
# tools/bench_detector.py --only-unet), C5 step (11 steps per run of --only-semi)
if [ -s $O/simsiam2d/out_kernel_stats.csv ]; then cp $O/simsiam2d/out_kernel_stats.csv profiles/${T}_simsiam2d_kernel_stats.csv; fi
if [ -d $O/pmc_simsiam2d ]; then python tools/pmc_summary.py $O/pmc_simsiam2d profiles/${T}_simsiam2d_traffic.json --sources conv_ --per-step 24 --kernels p2d_,conv_igemm,splitk,stem3_ \
  --note "tools/bench_simsiam2d.py --only-step (24 steps of batch 256): the conv family of the SimSiam-2D step" > /dev/null; fi
if [ -d $O/pmc_unet ]; then python tools/pmc_summary.py $O/pmc_unet profiles/${T}_unet_traffic.json --sources conv_,unet_ --per-step 5 \
  --note "tools/bench_detector.py --only-unet (5 forwards of unet_4 on a 128x512x512 tomogram; the loader's launches included)" > /dev/null; fi
if [ -d $O/pmc_c5 ]; then python tools/pmc_summary.py $O/pmc_c5 profiles/${T}_c5_traffic.json --sources loss_,conv_ --per-step 11 \
  --note "tools/bench_detector.py --only-semi (11 training steps of 16 pairs of 6x64x64)" > /dev/null; fi
if [ -f $O/graph/out_kernel_trace.csv ]; then python tools/trace_timeline.py $O/graph/out_kernel_trace.csv 2 > profiles/${T}_step_timeline.txt; fi
python - <<PY
import json
for f in ("conv_traffic", "mfma_busy", "infer_traffic"):
    d = json.load(open("profiles/${T}_%s.json" % f))
    print(f, d["source_sha16"], len(d["kernels"]), "kernels", d.get("hbm_bytes_per_step"))
PY
# tools/round_end.sh's other outputs, as they are
for pair in kernels_by_crop.txt:kernels_by_crop.txt detector.json:detector_bench.json stamp_plain.txt:stamp_step.txt stamp_dist.txt:stamp_step_dist.txt unet_layers.txt:unet_layers.txt; do
  src=$O/${pair%%:*}; dst=profiles/${T}_${pair##*:}
  if [ -s $src ]; then cp $src $dst; fi
done
