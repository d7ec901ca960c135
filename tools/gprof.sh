#!/bin/bash
# usage (on the GPU box): tools/gprof.sh <name> <script.py> [args...]   -> gpurun_out/<name>/ + a compact summary on stdout
# rocprofv3 --kernel-trace --stats with the python interpreter directly behind `--` (no wrapper hop).
# GPROF_FLAGS: extra rocprofv3 flags (e.g. --selected-regions with tools/prof_infer.py --regions)
R=${GRAFT_REPO_ROOT:-/root/repo}
name=$1; shift
mkdir -p $R/gpurun_out/$name
cd /tmp && export TMPDIR=/tmp
timeout -k 10 ${GPROF_TIMEOUT:-300} rocprofv3 --kernel-trace --stats ${GPROF_FLAGS} -d $R/gpurun_out/$name -o out --output-format csv -- python3 "$@" > $R/gpurun_out/$name/run.log 2>&1 || { tail -20 $R/gpurun_out/$name/run.log; exit 1; }
python3 $R/tools/stats_summary.py $R/gpurun_out/$name/out_kernel_stats.csv
