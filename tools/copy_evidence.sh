#!/bin/bash
# Copies the summaries of a tools/profile_round.sh run from gpurun_out/<tag>/ into profiles/<tag>_* (tracked):
#   bash tools/copy_evidence.sh r05
set -e
TAG=${1:-r05}
R=$(cd "$(dirname "$0")/.." && pwd)
S=$R/gpurun_out/$TAG
P=$R/profiles
cp $S/bench.json $P/${TAG}_bench.json
for f in kernel_stats kernel_stats_all_sections kernel_stats_fp16_cnn_64x64 kernel_stats_sampling kernel_stats_w16 kernel_stats_w32 kernel_stats_w32_fp16 \
         kernel_stats_w128 kernel_stats_w512 kernel_stats_w512_fp16 kernel_stats_train_w512 kernel_stats_train_w64 kernel_stats_train_w32_b138 kernel_stats_train_w32_b1024 pmc_fetch_counter_collection pmc_write_counter_collection \
         pmc_fp16_fetch_counter_collection pmc_fp16_write_counter_collection pmc_sampling_fetch_counter_collection pmc_sampling_write_counter_collection; do
  [ -f $S/$f.csv ] && cp $S/$f.csv $P/${TAG}_$f.csv
done
for f in sq_fp32_report sq_fp16_report sq_w32_report sq_w16_report sq_w32h_report sq_gemm512_report sq_gemm128_report sq_gemm512_fp16_report sq_gemm128_fp16_report gemm512_traffic train_w32_traffic sq_train32; do
  [ -f $S/$f.txt ] && cp $S/$f.txt $P/${TAG}_$f.txt
done
[ -f $S/mm_probe.log ] && cp $S/mm_probe.log $P/${TAG}_mm_probe.txt
[ -f $S/profile_map.md ] && cp $S/profile_map.md $P/${TAG}_profile_map.md
for f in sq_mm_l2fwd sq_mm_plain sq_mm_dw2; do [ -f $S/$f.txt ] && cp $S/$f.txt $P/${TAG}_$f.txt; done
cp $S/traffic.json $P/traffic.json
ls $P | grep "^${TAG}_"
