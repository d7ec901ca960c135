#!/bin/bash
# usage: tools/var_prof.sh <out-tag> "<python script + args>" "<kernel-name regex>" variant...
#   one rocprofv3 --kernel-trace --stats run per variant library (tools/var/lib_<variant>.so; "product" = the product library);
#   prints the matching rows of each run's kernel statistics
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
TAG=$1; CMD=$2; PAT=$3; shift 3
for V in "$@"; do
  O=gpurun_out/var/$TAG/$V
  rm -rf $O; mkdir -p $O
  if [ "$V" = product ]; then unset CETPICK_HIP_LIB; else export CETPICK_HIP_LIB=$PWD/tools/var/lib_$V.so; fi
  timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O -o out --output-format csv -- python3 $CMD > $O/stdout.log 2> $O/stderr.log
  echo "== $V"
  f=$(find $O -name "out_kernel_stats.csv" | head -1)
  if [ -n "$f" ]; then python3 - "$f" "$PAT" <<'PY'
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
pat = re.compile(sys.argv[2])
for r in rows:
    if pat.search(r["Name"]):
        print("  %-60s calls %5s  avg %9.2f us  min %9.2f  max %9.2f" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
PY
  else echo "  (no stats; see $O/stderr.log)"; tail -5 $O/stderr.log; fi
done
