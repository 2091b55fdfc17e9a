#!/bin/bash
# Tuning aid: an alternative build of ONE OR MORE objects of the library, linked with the product's other objects:
#   [OBJ="nf_kernels nf_host"] [SRC=path.hip] bash tools/build_variant.sh <name> [-DFLAG=..]...   ->  build/variants/lib_<name>.so
# OBJ names the object(s) that are replaced (default nf_kernels; nf_wide, nf_wide16, nf_gemm, nf_gemm16, nf_train, nf_host ...), SRC the
# source a single replaced object is compiled from (default csrc/$OBJ.hip; a patched copy must sit where its includes resolve, or be
# compiled with -I).  Use with NF_TOOL_LIB=build/variants/lib_<name>.so (tools/ only — the product loads csrc/libnoiseflow_hip.so).
set -e
NAME=$1; shift
R=$(cd "$(dirname "$0")/.." && pwd)
C=$R/noise_flow_amd/csrc
OBJ=${OBJ:-nf_kernels}
mkdir -p $R/build/variants
for ob in $OBJ; do
  EXTRA=""
  case $ob in nf_kernels|nf_host|nf_hostfed) ;; *) EXTRA="-mllvm -amdgpu-mfma-vgpr-form";; esac
  S=$C/$ob.hip
  [ -n "$SRC" ] && [ "$OBJ" = "$ob" ] && S=$SRC
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics -Wall -Wno-unused-function $EXTRA -I$C "$@" -c $S -o $R/build/variants/${ob}_$NAME.o &
done
wait
OBJS=""
for o in nf_kernels nf_wide nf_wide16 nf_gemm nf_gemm16 nf_host nf_hostfed nf_train; do
  if [[ " $OBJ " == *" $o "* ]]; then OBJS="$OBJS $R/build/variants/${o}_$NAME.o"; else OBJS="$OBJS $C/$o.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS -lpthread -o $R/build/variants/lib_$NAME.so
echo built $R/build/variants/lib_$NAME.so
