#!/bin/bash
# Copies the summaries of a tools/profile_round.sh run from gpurun_out/<tag>/ into profiles/<tag>_* (tracked):
#   bash tools/copy_evidence.sh r04
set -e
TAG=${1:-r04}
R=$(cd "$(dirname "$0")/.." && pwd)
S=$R/gpurun_out/$TAG
P=$R/profiles
cp $S/bench.json $P/${TAG}_bench.json
for f in kernel_stats kernel_stats_all_sections kernel_stats_fp16_cnn_64x64 kernel_stats_sampling pmc_fetch_counter_collection pmc_write_counter_collection \
         pmc_fp16_fetch_counter_collection pmc_fp16_write_counter_collection pmc_sampling_fetch_counter_collection pmc_sampling_write_counter_collection; do
  [ -f $S/$f.csv ] && cp $S/$f.csv $P/${TAG}_$f.csv
done
for f in sq_fp32_report sq_fp16_report sq_w32_report sq_w16_report sq_w32h_report sq_gemm512_report sq_gemm128_report sq_gemm512_fp16_report sq_gemm128_fp16_report gemm512_traffic; do
  [ -f $S/$f.txt ] && cp $S/$f.txt $P/${TAG}_$f.txt
done
cp $S/traffic.json $P/traffic.json
ls $P | grep "^${TAG}_"
