#!/bin/bash
# Timing-experiment builds of ONE kernel source (results may be garbage on purpose; never shipped):
#   tools/build_var.sh <file.hip> <name>=<flags> ...   ->  tools/var/lib_<name>.so   (select with CETPICK_HIP_LIB)
# The source compiled is tools/var/src/<file.hip> if it exists (an edited copy), else the product source.
# tools/var/ is git-ignored but travels to the GPU box.
set -e
cd "$(dirname "$0")/.."
SRC=$1; shift
python -m cet_pick_amd.build >/dev/null
mkdir -p tools/var
BASE=$(basename $SRC .hip)
FILE=cet_pick_amd/csrc/$SRC
EXTRA=$(grep -h "^// hipcc-flags:" $FILE | cut -d: -f2)
if [ -f tools/var/src/$SRC ]; then FILE=tools/var/src/$SRC; fi
OBJS=$(ls cet_pick_amd/csrc/_obj/*.o | grep -v "/$BASE\.o")
for V in "$@"; do
  NAME=${V%%=*}; FLAGS=${V#*=}
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $EXTRA $FLAGS -Icet_pick_amd/csrc -c $FILE -o /tmp/var_${BASE}_$NAME.o 2>/dev/null
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/var/lib_$NAME.so /tmp/var_${BASE}_$NAME.o $OBJS
  echo built tools/var/lib_$NAME.so
done
