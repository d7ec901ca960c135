#!/bin/bash
# HBM traffic of the train step's kernels: FETCH_SIZE and WRITE_SIZE in separate passes (MI355X_MICROARCH.md, HBM section)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/pmc_traffic
for C in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --pmc $C --kernel-trace -d $R/gpurun_out/pmc_traffic/$C -o out --output-format csv -- python3 $R/bench.py --no-secondary --no-cpu-baseline --steps 6 --warmup 2 --no-graph > $R/gpurun_out/pmc_traffic/$C.log 2>&1 || exit 1
done
