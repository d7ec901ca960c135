#!/bin/bash
# usage: tools/dbg_run.sh "<bench_conv args>" variant...
ARGS="$1"; shift
for V in "$@"; do
  echo "== $V"
  if [ "$V" = base ]; then timeout -k 10 120 python tools/bench_conv.py $ARGS 2>/dev/null
  else CETPICK_HIP_LIB=$PWD/tools/dbg/lib_$V.so timeout -k 10 120 python tools/bench_conv.py $ARGS 2>/dev/null; fi
done
