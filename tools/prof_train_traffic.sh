#!/bin/bash
# HBM traffic per trainer kernel:  bash tools/prof_train_traffic.sh <width> <batch>   (through gpurun; three passes, serial mode)
R=${GRAFT_REPO_ROOT:-$(pwd)}
W=${1:-32}; B=${2:-1024}
export NF_TRAIN_SERIAL=1
cd /tmp && export TMPDIR=/tmp
CMD="cd $R && python tools/bench_train_width.py $W $B"
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/tt_kt -- bash -c "$CMD" > $R/gpurun_out/tt_kt.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/tt_fetch -- bash -c "$CMD" > $R/gpurun_out/tt_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/tt_write -- bash -c "$CMD" > $R/gpurun_out/tt_write.log 2>&1
cd $R
F=$(find gpurun_out/tt_fetch -name "*counter_collection.csv" | head -1); Wr=$(find gpurun_out/tt_write -name "*counter_collection.csv" | head -1); K=$(find gpurun_out/tt_kt -name "*kernel_stats.csv" | head -1)
python tools/train_traffic.py $F $Wr $K | tee gpurun_out/train_traffic_w${W}_b${B}.txt
