#!/bin/bash
# usage (on the GPU box): tools/gbench.sh <name> [bench.py args]  -> gpurun_out/<name>.json/.err and a digest
R=${GRAFT_REPO_ROOT:-/root/repo}
name=$1; shift
mkdir -p $(dirname $R/gpurun_out/$name)
cd $R && timeout -k 10 ${GBENCH_TIMEOUT:-700} python bench.py "$@" > gpurun_out/$name.json 2> gpurun_out/$name.err || { tail -30 gpurun_out/$name.err; exit 1; }
tail -2 gpurun_out/$name.err
python tools/bench_digest.py gpurun_out/$name.json
