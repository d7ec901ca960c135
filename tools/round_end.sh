#!/bin/bash
# Everything measured that goes under profiles/ at the end of a round (run on the GPU box through gpurun):
#   tools/round_end.sh <tag>     -> raw output under gpurun_out/<tag>_final/, summaries by tools/refresh_profiles_local.sh <tag>
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
T=${1:-r06}
O=$R/gpurun_out/${T}_final
cd $R
tools/refresh_profiles.sh $T
( echo "python tools/bench_conv.py --kernels <crop> <batch>: the kernel family each convolution of the MoCo-3D encoder takes and its time"
  echo "(us per call, hipGraph replay of 4 calls; direct3 / direct3s / s2 rows include the eager path's weight-image launch, which the engine"
  echo "does once per step instead)."
  echo
  for c in "32 64" "64 32" "64 8" "48 16"; do timeout -k 10 200 python tools/bench_conv.py --kernels $c 2>/dev/null; echo; done ) > $O/kernels_by_crop.txt
timeout -k 10 400 python tools/bench_detector.py > $O/detector.json 2> $O/detector.err
# rocprofv3 kernel summary of the C5 step alone
GPROF_TIMEOUT=200 tools/gprof.sh ${T}_final/c5 $R/tools/bench_detector.py --only-semi > $O/c5.txt 2>&1 || true
timeout -k 10 300 python tools/ab/unet_layers.py > $O/unet_layers.txt 2> $O/unet_layers.err
timeout -k 10 200 python tools/stamp_step.py 64 50 > $O/stamp_plain.txt 2>&1
timeout -k 10 200 python tools/stamp_step.py 64 50 --dist > $O/stamp_dist.txt 2>&1
echo "round_end done"
