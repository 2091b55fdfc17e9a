R=$PWD; O=$R/gpurun_out/sq32; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
SQ1="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_INSTS_SALU GRBM_GUI_ACTIVE"
SQ2="SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE"
NF_TOOL_STEPS=3 rocprofv3 --pmc $SQ1 --kernel-trace --output-format csv -d $O/p1 -- python $R/tools/bench_train_width.py 32 1024 > $O/p1.log 2>&1
NF_TOOL_STEPS=3 rocprofv3 --pmc $SQ2 --kernel-trace --output-format csv -d $O/p2 -- python $R/tools/bench_train_width.py 32 1024 > $O/p2.log 2>&1
cd $R
for k in k_c3_fwd_mfma k_c3_dh_mfma k_c2_bwd_mfma k_w1_grad_mfma k_c1_dz_mfma; do echo "== $k"; python tools/pmc_report.py $O $k | grep "^#"; done
