#!/bin/bash
# One captured training step as a kernel timeline (run on the GPU box):  tools/timeline_run.sh <out dir under gpurun_out>
# -> gpurun_out/<dir>/timeline.txt (tools/trace_timeline.py on a rocprofv3 kernel trace of bench.py's captured step)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/${1:-timeline}
mkdir -p $O
( cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace -d $O -o out --output-format csv -- python3 $R/bench.py --no-conv-profile --no-secondary --no-cpu-baseline --no-entry-point --steps 12 --warmup 5 > $O/run.log 2>&1 )
python3 $R/tools/trace_timeline.py $O/out_kernel_trace.csv 2 > $O/timeline.txt
tail -3 $O/timeline.txt
