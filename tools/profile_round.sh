#!/bin/bash
# Round evidence on the GPU box (run through gpurun):  bash tools/profile_round.sh <tag>   e.g. r02
# Writes everything under gpurun_out/<tag>/; copy the summaries you want judged into profiles/.
# Every counter pass is its own run with --kernel-trace only (never combined with other trace domains).
set -u
TAG=${1:-r05}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python $R/bench.py --steps 20 --warmup 5"
if [ "${2:-all}" = "traffic" ]; then
  # bash tools/profile_round.sh <tag> traffic: only the HBM-traffic passes of the headline kernel + the bench line that quotes
  # them (after a change to the kernel sources that cannot have changed the other evidence, e.g. a comment)
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- $BENCH --no-cpu-baseline --no-extras --ramp-ms 0 > $OUT/pmc_fetch.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- $BENCH --no-cpu-baseline --no-extras --ramp-ms 0 > $OUT/pmc_write.log 2>&1
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- $BENCH --no-cpu-baseline --no-extras > $OUT/kt.log 2>&1
  cd $R
  F=$(find $OUT/pmc_fetch -name "*counter_collection.csv" | head -1); W=$(find $OUT/pmc_write -name "*counter_collection.csv" | head -1)
  python tools/make_traffic.py $F $W > $OUT/traffic.log 2>&1 && cp profiles/traffic.json $OUT/traffic.json
  cp $F $OUT/pmc_fetch_counter_collection.csv; cp $W $OUT/pmc_write_counter_collection.csv
  K=$(find $OUT/kt -name "*kernel_stats.csv" | head -1); cp $K $OUT/kernel_stats.csv 2>/dev/null
  $BENCH > $OUT/bench.json 2> $OUT/bench.err
  tail -5 $OUT/traffic.log; head -3 $OUT/kernel_stats.csv; head -c 400 $OUT/bench.json; echo
  exit 0
fi
# 1. the driver's command, plain (the JSON line) and under the kernel trace
$BENCH > $OUT/bench.json 2> $OUT/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- $BENCH --no-cpu-baseline --no-extras > $OUT/kt.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_full -- $BENCH --no-cpu-baseline > $OUT/kt_full.log 2>&1
# 2. HBM traffic of the headline kernel: FETCH_SIZE and WRITE_SIZE in separate passes (headline only: --no-extras)
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- $BENCH --no-cpu-baseline --no-extras --ramp-ms 0 > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- $BENCH --no-cpu-baseline --no-extras --ramp-ms 0 > $OUT/pmc_write.log 2>&1
# 2b. HBM traffic of the other measured BASELINE configs: configs[4] shape (64x64, fp16 CNN, B = 1024) and configs[2] (sampling, B = 4096)
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc_fp16_$c -- python $R/tools/prof_nll.py 1024 20 64 fp16 > $OUT/pmc_fp16_$c.log 2>&1
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc_samp_$c -- python $R/tools/prof_nll.py 4096 20 32 fp32 sample > $OUT/pmc_samp_$c.log 2>&1
done
# per-config kernel stats
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_fp16 -- python $R/tools/prof_nll.py 1024 200 64 fp16 > $OUT/kt_fp16.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_samp -- python $R/tools/prof_nll.py 4096 100 32 fp32 sample > $OUT/kt_samp.log 2>&1
# per-kernel stats of bench.py's wide_cnn section (same model / batch, warm clocks): w16, w32, w32_fp16, w128, w512, w512_fp16
for spec in "16 fp32 w16" "32 fp32 w32" "32 fp16 w32_fp16" "128 fp32 w128" "512 fp32 w512" "512 fp16 w512_fp16"; do
  set -- $spec
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_$3 -- python $R/tools/prof_wide.py $1 $2 > $OUT/kt_$3.log 2>&1
  K=$(find $OUT/kt_$3 -name "*kernel_stats.csv" | head -1); cp $K $OUT/kernel_stats_$3.csv 2>/dev/null
done
set -- $TAG
# the training step at the reference's default width 512 and at 64 (csrc/nf_train_mm.h: this repo's own GEMMs)
for w in 512 64; do
  NF_TOOL_STEPS=10 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_train$w -- python $R/tools/bench_train_width.py $w 138 > $OUT/kt_train$w.log 2>&1
  K=$(find $OUT/kt_train$w -name "*kernel_stats.csv" | head -1); cp $K $OUT/kernel_stats_train_w$w.csv 2>/dev/null
done
# the width-32 step on the patch-resident stages (csrc/nf_train_pr.h): per-kernel durations in serial mode at 138 and 1 024 patches,
# HBM bytes per kernel and SQ counters at 1 024
for b in 138 1024; do
  NF_TRAIN_SERIAL=1 NF_TOOL_STEPS=10 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_train32_$b -- python $R/tools/bench_train_width.py 32 $b > $OUT/kt_train32_$b.log 2>&1
  K=$(find $OUT/kt_train32_$b -name "*kernel_stats.csv" | head -1); cp $K $OUT/kernel_stats_train_w32_b$b.csv 2>/dev/null
done
(cd $R && bash tools/prof_train_traffic.sh 32 1024 > $OUT/train_w32_traffic.txt 2>&1)
(cd $R && bash tools/sq_train32.sh 1024 > $OUT/sq_train32.txt 2>&1)
cd /tmp
(timeout 300 $R/tools/probes/mm_probe time > $OUT/mm_probe.log 2>&1)
# SQ counters of the trainer's three GEMM shapes at width 512 (probe products 0 = l_2 forward, 2 = plain, 8 = d l_2/W)
SQA="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_INSTS_SALU GRBM_GUI_ACTIVE"
SQB="SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE"
for spec in "0 l2fwd" "2 plain" "8 dw2"; do
  set -- $spec
  rocprofv3 --pmc $SQA --kernel-trace --output-format csv -d $OUT/sq_mm_$2/p1 -- $R/tools/probes/mm_probe time $1 512 > /dev/null 2>&1
  rocprofv3 --pmc $SQB --kernel-trace --output-format csv -d $OUT/sq_mm_$2/p2 -- $R/tools/probes/mm_probe time $1 512 > /dev/null 2>&1
  (cd $R && python tools/pmc_report.py $OUT/sq_mm_$2 k_mm_ > $OUT/sq_mm_$2.txt 2>&1)
done
set -- $TAG
# 3. SQ counters: headline kernel at B = 16384, fp16-CNN 64x64, wide CNN width 32
SQ1="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_INSTS_SALU GRBM_GUI_ACTIVE"
SQ2="SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE"
for pass in 1 2; do
  eval C=\$SQ$pass
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/sq_fp32/p$pass -- python $R/tools/prof_nll.py 16384 4 32 fp32 > $OUT/sq_fp32_p$pass.log 2>&1
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/sq_fp16/p$pass -- python $R/tools/prof_nll.py 2048 4 64 fp16 > $OUT/sq_fp16_p$pass.log 2>&1
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/sq_w32/p$pass -- python $R/tools/quick_time_wide.py 32 8192 32 4 > $OUT/sq_w32_p$pass.log 2>&1
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/sq_w16/p$pass -- python $R/tools/quick_time_wide.py 16 8192 32 4 > $OUT/sq_w16_p$pass.log 2>&1
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/sq_w32h/p$pass -- python $R/tools/quick_time_wide.py 32 8192 32 4 fp16 > $OUT/sq_w32h_p$pass.log 2>&1
  B=512 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/sq_gemm512/p$pass -- python $R/tools/check_gemm.py 512 > $OUT/sq_gemm512_p$pass.log 2>&1
  B=512 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/sq_gemm128/p$pass -- python $R/tools/check_gemm.py 128 > $OUT/sq_gemm128_p$pass.log 2>&1
  B=512 DTYPE=fp16 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/sq_gemm512h/p$pass -- python $R/tools/check_gemm.py 512 > $OUT/sq_gemm512h_p$pass.log 2>&1
  B=512 DTYPE=fp16 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/sq_gemm128h/p$pass -- python $R/tools/check_gemm.py 128 > $OUT/sq_gemm128h_p$pass.log 2>&1
done
# HBM traffic of the GEMM kernel at width 512 (weights are re-read per band from L2, not from HBM)
B=512 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/gemm512_fetch -- python $R/tools/check_gemm.py 512 > $OUT/gemm512_fetch.log 2>&1
B=512 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/gemm512_write -- python $R/tools/check_gemm.py 512 > $OUT/gemm512_write.log 2>&1
cd $R
python tools/pmc_report.py $OUT/sq_fp32 "nf_flow_kernel<4, 256, 4, false, true, true, 0, false>" 100000 > $OUT/sq_fp32_report.txt 2>&1
python tools/pmc_report.py $OUT/sq_fp16 "nf_flow_kernel<4, 1024, 4, false, true, true, 2, false>" 100000 > $OUT/sq_fp16_report.txt 2>&1
python tools/pmc_report.py $OUT/sq_w32 "nf_wide32_kernel" 100000 > $OUT/sq_w32_report.txt 2>&1
python tools/pmc_report.py $OUT/sq_w16 "nf_wide16_kernel" 100000 > $OUT/sq_w16_report.txt 2>&1
python tools/pmc_report.py $OUT/sq_w32h "nf_wide32_kernel" 100000 > $OUT/sq_w32h_report.txt 2>&1
python tools/pmc_report.py $OUT/sq_gemm512 "nf_gemm_kernel<512" 100000 > $OUT/sq_gemm512_report.txt 2>&1
python tools/pmc_report.py $OUT/sq_gemm128 "nf_gemmb_kernel<128" 100000 > $OUT/sq_gemm128_report.txt 2>&1
python tools/pmc_report.py $OUT/sq_gemm512h "nf_gemm16_kernel<512" 100000 > $OUT/sq_gemm512_fp16_report.txt 2>&1
python tools/pmc_report.py $OUT/sq_gemm128h "nf_gemm16b_kernel<128" 100000 > $OUT/sq_gemm128_fp16_report.txt 2>&1
python tools/pmc_report.py $OUT/gemm512_fetch "nf_gemm_kernel<512" 100000 > $OUT/gemm512_traffic.txt 2>&1
python tools/pmc_report.py $OUT/gemm512_write "nf_gemm_kernel<512" 100000 >> $OUT/gemm512_traffic.txt 2>&1
F=$(find $OUT/pmc_fetch -name "*counter_collection.csv" | head -1); W=$(find $OUT/pmc_write -name "*counter_collection.csv" | head -1)
python tools/make_traffic.py $F $W > $OUT/traffic.log 2>&1
F2=$(find $OUT/pmc_fp16_FETCH_SIZE -name "*counter_collection.csv" | head -1); W2=$(find $OUT/pmc_fp16_WRITE_SIZE -name "*counter_collection.csv" | head -1)
python tools/make_traffic.py $F2 $W2 "nf_flow_kernel<4, 1024, 4, false, true, true, 2, false>" 1024 fp16_cnn_64x64 1024 131072 >> $OUT/traffic.log 2>&1
F3=$(find $OUT/pmc_samp_FETCH_SIZE -name "*counter_collection.csv" | head -1); W3=$(find $OUT/pmc_samp_WRITE_SIZE -name "*counter_collection.csv" | head -1)
python tools/make_traffic.py $F3 $W3 "nf_flow_kernel<4, 256, 4, true, true, true, 0, false>" 4096 sampling 256 32768 >> $OUT/traffic.log 2>&1
cp profiles/traffic.json $OUT/traffic.json
# the bench line once more, now that profiles/traffic.json belongs to these kernel sources (roofline.traffic is quoted from it)
mv $OUT/bench.json $OUT/bench_before_traffic.json
(cd /tmp && $BENCH > $OUT/bench.json 2> $OUT/bench.err)
cp $F2 $OUT/pmc_fp16_fetch_counter_collection.csv; cp $W2 $OUT/pmc_fp16_write_counter_collection.csv
cp $F3 $OUT/pmc_sampling_fetch_counter_collection.csv; cp $W3 $OUT/pmc_sampling_write_counter_collection.csv
K=$(find $OUT/kt -name "*kernel_stats.csv" | head -1); cp $K $OUT/kernel_stats.csv 2>/dev/null
K=$(find $OUT/kt_fp16 -name "*kernel_stats.csv" | head -1); cp $K $OUT/kernel_stats_fp16_cnn_64x64.csv 2>/dev/null
K=$(find $OUT/kt_samp -name "*kernel_stats.csv" | head -1); cp $K $OUT/kernel_stats_sampling.csv 2>/dev/null
K=$(find $OUT/kt_full -name "*kernel_stats.csv" | head -1); cp $K $OUT/kernel_stats_all_sections.csv 2>/dev/null
cp $F $OUT/pmc_fetch_counter_collection.csv; cp $W $OUT/pmc_write_counter_collection.csv
for f in $OUT/sq_fp32_report.txt $OUT/sq_fp16_report.txt $OUT/sq_w32_report.txt $OUT/sq_w16_report.txt $OUT/sq_w32h_report.txt $OUT/sq_gemm512_report.txt $OUT/sq_gemm128_report.txt $OUT/sq_gemm512_fp16_report.txt $OUT/sq_gemm128_fp16_report.txt $OUT/gemm512_traffic.txt; do tail -n 3 $f; done; head -c 600 $OUT/bench.json; echo; tail -5 $OUT/traffic.log; head -8 $OUT/kernel_stats.csv
python tools/check_profile_map.py $OUT/bench.json $OUT "" > $OUT/profile_map.md 2>&1; tail -12 $OUT/profile_map.md
