#!/bin/bash
# Timing-experiment builds of the conv kernels (tools/dbg/lib_<name>.so; select with CETPICK_HIP_LIB).  The product
# sources carry no experiment code: the variants (MI_DBG_NOGLOAD, MI_DBG_BF3_NOCVT, MI_DBG_NOSTORE, MI_DBG_BF3_ONEPROD,
# MI_DBG_NOBARRIER, MI_DBG_STEM_NOPATCH, MI_DBG_STEM_ONEPROD - each one computes garbage on purpose) live in
# tools/conv_dbg_variants.patch, applied here to a COPY of the sources.
#   tools/build_dbg.sh BF3_NOCVT BF3_ONEPROD BF3_NOCVT,BF3_ONEPROD
set -e
cd "$(dirname "$0")/.."
python -m cet_pick_amd.build >/dev/null
rm -rf tools/dbg/src && mkdir -p tools/dbg/src/cet_pick_amd tools/dbg/src/include
cp -r cet_pick_amd/csrc tools/dbg/src/cet_pick_amd/ && cp include/*.h tools/dbg/src/include/
(cd tools/dbg/src && patch -p1 -s < ../../conv_dbg_variants.patch)
OBJS=$(ls cet_pick_amd/csrc/_obj/*.o | grep -v "conv_igemm\|conv_stem")
for V in "$@"; do
  D=""; for m in $(echo $V | tr ',' ' '); do D="$D -DMI_DBG_$m"; done
  for f in conv_igemm conv_stem; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $D -c tools/dbg/src/cet_pick_amd/csrc/$f.hip -o tools/dbg/${f}_$V.o 2>/dev/null
  done
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/dbg/lib_$V.so tools/dbg/conv_igemm_$V.o tools/dbg/conv_stem_$V.o $OBJS
done
