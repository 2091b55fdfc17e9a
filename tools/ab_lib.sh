#!/bin/bash
# Alternates builds of the library (build/variants/lib_<name>.so from tools/build_variant.sh; "product" = csrc/libnoiseflow_hip.so)
# on the forward timings, ONE box:   gpurun -- 'LIBS="base product" bash tools/ab_lib.sh'      [M=fp16,fp32] [CFGS="64 1024 400;32 1024 2000"]
R=$(cd "$(dirname "$0")/.." && pwd)
LIBS=${LIBS:-"base product"}
IFS=';' read -ra CF <<< "${CFGS:-64 1024 400;32 1024 2000}"
for rep in 1 2 3; do
  for lib in $LIBS; do
    L=""; [ $lib != product ] && L=$R/build/variants/lib_$lib.so
    for cfg in "${CF[@]}"; do
      echo "$lib $cfg: $(NF_TOOL_LIB=$L MODES=${M:-fp16,fp32} python $R/tools/quick_time_fp16.py $cfg 2>/dev/null | grep nll | tr '\n' ' ')"
    done
  done
done
