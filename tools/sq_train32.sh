# SQ counter passes over the width-32 training step, one report per stage kernel:  bash tools/sq_train32.sh [B] [kernels...]
B=${1:-1024}; shift
KS=${@:-"k_pr_fwd<1 k_pr_fwd<2 k_pr_bwd<0 k_pr_bwd<1 k_pr_bwd<2"}
R=$PWD; O=$R/gpurun_out/sq32; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
SQ1="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_INSTS_SALU GRBM_GUI_ACTIVE"
SQ2="SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE"
NF_TOOL_STEPS=3 rocprofv3 --pmc $SQ1 --kernel-trace --output-format csv -d $O/p1 -- python $R/tools/bench_train_width.py 32 $B > $O/p1.log 2>&1
NF_TOOL_STEPS=3 rocprofv3 --pmc $SQ2 --kernel-trace --output-format csv -d $O/p2 -- python $R/tools/bench_train_width.py 32 $B > $O/p2.log 2>&1
cd $R
for k in $KS; do echo "== $k"; python tools/pmc_report.py $O "$k" | grep "^#"; done
