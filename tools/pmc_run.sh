#!/bin/bash
# usage (on the GPU box): tools/pmc_run.sh <name> <counter-set> <script.py> [args...]
# One rocprofv3 --pmc pass per counter GROUP (groups separated by ':', counters of a group by ','), each with
# --kernel-trace only (MI355X_MICROARCH.md: FETCH_SIZE and WRITE_SIZE do not fit one pass; no other trace domain is
# combined with --pmc).  The python interpreter sits directly behind `--`.
R=${GRAFT_REPO_ROOT:-/root/repo}
name=$1; sets=$2; shift 2
cd /tmp && export TMPDIR=/tmp
IFS=':' read -ra groups <<< "$sets"
for g in "${groups[@]}"; do
  tag=${g//,/+}
  mkdir -p $R/gpurun_out/$name/$tag
  timeout -k 10 ${PMC_TIMEOUT:-300} rocprofv3 --pmc ${g//,/ } --kernel-trace -d $R/gpurun_out/$name/$tag -o out --output-format csv -- python3 "$@" > $R/gpurun_out/$name/$tag/run.log 2>&1 || { tail -20 $R/gpurun_out/$name/$tag/run.log; exit 1; }
  echo "pass $tag done"
done
