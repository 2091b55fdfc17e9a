#!/bin/bash
# Assembly of the kernels under study only (-DNF_ISA_PROBE: seconds):  bash tools/isa_probe.sh out.s [-DFLAG=..]...
# then  python tools/isa_budget.py out.s '<kernel name>' [labels]
set -e
OUT=$1; shift
C=$(cd "$(dirname "$0")/../noise_flow_amd/csrc" && pwd)
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -munsafe-fp-atomics --cuda-device-only -DNF_ISA_PROBE "$@" -S $C/nf_kernels.hip -o $OUT 2>&1 | grep -v "hip-link" || true
