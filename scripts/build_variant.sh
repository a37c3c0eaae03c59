#!/bin/bash
# A/B builds of the library with extra compile flags: scripts/build_variant.sh NAME "-DFOO=1 ..."
# -> hevc-complexity-reduction_amd/lib_NAME/libethcnn.so ; run with ETHCNN_LIB=<that path>
# extra make variables through MAKEVARS, e.g. MAKEVARS="KFLAGS_DENSE=" (FC1 accumulators in AGPRs)
set -eu
NAME=$1; shift
cd "$(dirname "$0")/../hevc-complexity-reduction_amd/csrc"
make -s -j8 OUTDIR=../lib_$NAME OBJDIR=_obj_$NAME XFLAGS="$*" ${MAKEVARS:-} ../lib_$NAME/libethcnn.so
echo "built ../lib_$NAME/libethcnn.so"
