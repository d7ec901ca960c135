#!/bin/bash
# timing-experiment builds of the stem kernels (tools/dbg/lib_<name>.so); select with CETPICK_HIP_LIB
set -e
cd "$(dirname "$0")/.."
python -m cet_pick_amd.build >/dev/null
mkdir -p tools/dbg
OBJS=$(ls cet_pick_amd/csrc/_obj/*.o | grep -v conv_stem)
for V in "$@"; do
  D=""; for m in $(echo $V | tr ',' ' '); do D="$D -DMI_DBG_$m"; done
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $D -c cet_pick_amd/csrc/conv_stem.hip -o tools/dbg/stem_$V.o 2>/dev/null
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/dbg/lib_$V.so tools/dbg/stem_$V.o $OBJS
done
