#!/bin/bash
# A/B of the two width-4 fp16-CNN formulations on ONE box (run through gpurun):  bash tools/ab_fp16.sh [tag]
#   NF_H16=4x4  v_mfma_f32_4x4x4_16b_f16 (2-pass, owns the issue port)      default  v_mfma_f32_16x16x32_f16 (NF11_*)
# parity tests of the mode first, then alternating timings, then the SQ counter passes of the default kernel.
set -u
TAG=${1:-h16}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "fp16" > $OUT/tests.log 2>&1; echo "tests rc=$?"; tail -4 $OUT/tests.log
for rep in 1 2; do
  for h16 in 16x16 4x4; do
    for cfg in "64 1024 300" "32 16384 60" "32 1024 600"; do
      echo "NF_H16=$h16 $cfg: $(NF_H16=$h16 MODES=fp16 python tools/quick_time_fp16.py $cfg | tr '\n' ' ')"
    done
  done
done 2>&1 | tee $OUT/ab.log
cd /tmp && export TMPDIR=/tmp
SQ1="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_INSTS_SALU GRBM_GUI_ACTIVE"
SQ2="SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE"
for pass in 1 2; do
  eval C=\$SQ$pass
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/sq_fp16/p$pass -- python $R/tools/prof_nll.py 2048 4 64 fp16 > $OUT/sq_fp16_p$pass.log 2>&1
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/sq_fp16_32/p$pass -- python $R/tools/prof_nll.py 16384 4 32 fp16 > $OUT/sq_fp16_32_p$pass.log 2>&1
done
cd $R
python tools/pmc_report.py $OUT/sq_fp16 "nf_flow_kernel<4, 1024, 4, false, true, true, 2, false>" 100000 > $OUT/sq_fp16_report.txt 2>&1
python tools/pmc_report.py $OUT/sq_fp16_32 "nf_flow_kernel<4, 256, 4, false, true, true, 2, false>" 100000 > $OUT/sq_fp16_32_report.txt 2>&1
tail -n 9 $OUT/sq_fp16_report.txt $OUT/sq_fp16_32_report.txt
