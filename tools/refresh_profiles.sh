#!/bin/bash
# Round-end collection of everything under profiles/ that is measured (run on the GPU box through gpurun):
#   tools/refresh_profiles.sh <round tag, e.g. r02>
# leaves raw rocprofv3 output under gpurun_out/<tag>_final/; tools/refresh_profiles_local.sh turns it into profiles/<tag>_*.
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
T=${1:-r06}
O=$T"_final"
mkdir -p $R/gpurun_out/$O
cd $R
# 1. kernel traces (rocprofv3 --kernel-trace --stats, program directly behind `--`)
tools/gprof.sh $O/train $R/bench.py --no-graph --no-conv-profile --no-secondary --no-cpu-baseline --no-entry-point --steps 100 --warmup 10 > gpurun_out/$O/train.txt 2>&1
tools/gprof.sh $O/infer $R/tools/prof_infer.py both 18 > gpurun_out/$O/infer.txt 2>&1
# 1b. kernel trace of the CAPTURED step (tools/trace_timeline.py: one replay as a timeline)
mkdir -p $R/gpurun_out/$O/graph
( cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace -d $R/gpurun_out/$O/graph -o out --output-format csv -- python3 $R/bench.py --no-conv-profile --no-secondary --no-cpu-baseline --no-entry-point --steps 12 --warmup 5 > $R/gpurun_out/$O/graph/run.log 2>&1 )
echo "traces done"
# 2. PMC passes (counters only, one pass per group)
tools/pmc_run.sh $O/pmc_infer FETCH_SIZE:WRITE_SIZE $R/tools/prof_infer.py both 3
tools/pmc_run.sh $O/pmc_train FETCH_SIZE:WRITE_SIZE:SQ_VALU_MFMA_BUSY_CYCLES,GRBM_GUI_ACTIVE $R/bench.py --no-graph --no-conv-profile --no-secondary --no-cpu-baseline --no-entry-point --steps 6 --warmup 3
# round 6: traffic of the SimSiam-2D step, the unet_4 forward and the C5 step from their own PMC passes
tools/gprof.sh $O/simsiam2d $R/tools/bench_simsiam2d.py --only-step > gpurun_out/$O/simsiam2d.txt 2>&1
tools/pmc_run.sh $O/pmc_simsiam2d FETCH_SIZE:WRITE_SIZE $R/tools/bench_simsiam2d.py --only-step --no-graph
tools/pmc_run.sh $O/pmc_unet FETCH_SIZE:WRITE_SIZE $R/tools/bench_detector.py --only-unet
PMC_TIMEOUT=400 tools/pmc_run.sh $O/pmc_c5 FETCH_SIZE:WRITE_SIZE $R/tools/bench_detector.py --only-semi
echo "pmc done"
