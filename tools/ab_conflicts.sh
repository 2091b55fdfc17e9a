#!/bin/bash
# Tuning aid: LDS bank-conflict cycles of the 64x64 fp16-CNN kernel for library variants:  bash tools/ab_conflicts.sh name1 name2 ...
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/conf; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  NF_TOOL_LIB=$R/build/variants/lib_$v.so rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $OUT/$v -- python $R/tools/prof_nll.py 2048 4 64 fp16 > $OUT/$v.log 2>&1
  echo "== $v"; python $R/tools/pmc_report.py $OUT/$v "nf_flow_kernel<4, 1024, 4, false, true, true, 2" 100000 | grep -E "^SQ|median"
done
