#!/bin/bash
# per-kernel durations of the training step at a coupling width (serial mode: no overlap):  bash tools/prof_train_width.sh 32 138 [tag]
R=${GRAFT_REPO_ROOT:-$(pwd)}
W=${1:-32}; B=${2:-138}; TAG=${3:-trw}
cd /tmp && export TMPDIR=/tmp
NF_TRAIN_SERIAL=1 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$TAG -- bash -c "cd $R && python tools/bench_train_width.py $W $B" > $R/gpurun_out/$TAG.log 2>&1
cd $R
f=$(find gpurun_out/$TAG -name "*kernel_stats.csv" | head -1)
python - <<PY
import csv
rows=list(csv.DictReader(open("$f")))
for r in rows[:20]:
    name=r["Name"].replace("void ","").replace("(anonymous namespace)::","").split("(")[0][:44]
    print("%-46s calls %5s avg %8.1f us  total %8.1f ms" % (name, r["Calls"], float(r["AverageNs"])/1e3, float(r["TotalDurationNs"])/1e6))
PY
