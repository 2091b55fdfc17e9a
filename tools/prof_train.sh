#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/trprof -- bash -c "cd $R && python tools/bench_train.py --batches ${1:-138} --steps 20" > $R/gpurun_out/trprof.log 2>&1
cd $R
f=$(find gpurun_out/trprof -name "*kernel_stats.csv" | head -1)
python - <<PY
import csv
rows=list(csv.DictReader(open("$f")))
for r in rows[:24]:
    name=r["Name"].replace("void ","").replace("(anonymous namespace)::","").split("(")[0][:44]
    print("%-46s calls %5s avg %8.1f us  total %8.1f ms" % (name, r["Calls"], float(r["AverageNs"])/1e3, float(r["TotalDurationNs"])/1e6))
PY
