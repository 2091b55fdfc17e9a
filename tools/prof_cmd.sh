#!/bin/bash
# kernel-trace summary of any command:  bash tools/prof_cmd.sh <outdir-name> <command...>   (run through gpurun)
R=${GRAFT_REPO_ROOT:-$(pwd)}
D=$1; shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$D -- bash -c "cd $R && $*" > $R/gpurun_out/$D.log 2>&1
cd $R
f=$(find gpurun_out/$D -name "*kernel_stats.csv" | head -1)
python - <<PY
import csv
rows=list(csv.DictReader(open("$f")))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:26]:
    name=r["Name"].replace("void ","").replace("(anonymous namespace)::","").split("(")[0][:48]
    print("%-50s calls %5s avg %9.1f us  total %8.1f ms  %5.1f %%" % (name, r["Calls"], float(r["AverageNs"])/1e3, float(r["TotalDurationNs"])/1e6, 100*float(r["TotalDurationNs"])/tot))
PY
