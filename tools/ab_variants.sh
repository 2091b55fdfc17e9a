#!/bin/bash
# Same-box A/B of library variants (tools/build_variant.sh) on the fp16-CNN mode:  bash tools/ab_variants.sh "<cfg>" name1 name2 ...
# cfg = "H B iters" for tools/quick_time_fp16.py; MODES (default fp16) as there.  Two alternating rounds.
R=${GRAFT_REPO_ROOT:-$(pwd)}
CFG=$1; shift
for rep in 1 2; do
  for v in "$@"; do
    echo "$v [$CFG]: $(NF_TOOL_LIB=$R/build/variants/lib_$v.so MODES=${MODES:-fp16} python $R/tools/quick_time_fp16.py $CFG 2>/dev/null | tr '\n' ' ')"
  done
done
