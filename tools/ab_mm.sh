#!/bin/bash
# Timing ablations of the trainer's k_mm_pix (MM_ABL bits of csrc/nf_train_mm.h) on ONE box:
#   gpurun -- 'bash tools/ab_mm.sh "0 1 2 8 16 32" "0 2"'      (ablation values, product indices of `mm_probe time`)
R=$(cd "$(dirname "$0")/.." && pwd)
for abl in $1; do
  hipcc -O3 -std=c++17 --offload-arch=gfx950 -munsafe-fp-atomics -mllvm -amdgpu-mfma-vgpr-form -DMM_ABL=$abl $MM_EXTRA -I$R/noise_flow_amd/csrc \
        $R/tools/probes/mm_probe.hip -o /tmp/mm_probe_$abl &
done
wait
for rep in 1 2; do
  for abl in $1; do
    for k in $2; do echo -n "ABL=$abl "; /tmp/mm_probe_$abl time $k ${3:-512} | grep "^time"; done
  done
done
