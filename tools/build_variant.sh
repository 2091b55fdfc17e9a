#!/bin/bash
# Tuning aid: an alternative build of nf_kernels.hip linked with the other objects of the library:
#   bash tools/build_variant.sh <name> [-DFLAG=..]...   ->  build/variants/lib_<name>.so   (use with NF_TOOL_LIB=...)
set -e
NAME=$1; shift
R=$(cd "$(dirname "$0")/.." && pwd)
C=$R/noise_flow_amd/csrc
mkdir -p $R/build/variants
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics -Wall -Wno-unused-function "$@" -c ${SRC:-$C/nf_kernels.hip} -o $R/build/variants/nf_kernels_$NAME.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $R/build/variants/nf_kernels_$NAME.o $C/nf_wide.o $C/nf_wide16.o $C/nf_gemm.o $C/nf_gemm16.o $C/nf_host.o $C/nf_hostfed.o $C/nf_train.o -lpthread -o $R/build/variants/lib_$NAME.so
echo built $R/build/variants/lib_$NAME.so
