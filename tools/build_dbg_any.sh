#!/bin/bash
# experiment builds of ONE source file with extra -D flags: tools/build_dbg_any.sh <file.hip> <name>=<flags> ...
# -> tools/dbg/lib_<name>.so (select with CETPICK_HIP_LIB)
set -e
cd "$(dirname "$0")/.."
SRC=$1; shift
python -m cet_pick_amd.build >/dev/null
mkdir -p tools/dbg
BASE=$(basename $SRC .hip)
OBJS=$(ls cet_pick_amd/csrc/_obj/*.o | grep -v "/$BASE\.")
for V in "$@"; do
  NAME=${V%%=*}; FLAGS=${V#*=}
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $FLAGS -c cet_pick_amd/csrc/$SRC -o tools/dbg/${BASE}_$NAME.o 2>/dev/null
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/dbg/lib_$NAME.so tools/dbg/${BASE}_$NAME.o $OBJS
done
