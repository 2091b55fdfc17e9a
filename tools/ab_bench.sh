#!/bin/bash
# A/B aid: the headline leg of bench.py through a given library build.  tools/ab_bench.sh lib.so [bench args]
lib=$(realpath "$1"); shift
python - "$lib" "$@" <<'PY' 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('value %.4g  ms_per_step %.5f  kernel_ms %.5f' % (d['value'], d['ms_per_step'], d['roofline']['kernel_ms']))"
import os, runpy, sys
root = os.getcwd()
sys.path.insert(0, root)
from noise_flow_amd import _lib
_lib.LIB_PATH = sys.argv[1]
sys.argv = ["bench.py", "--no-extras", "--no-cpu-baseline"] + sys.argv[2:]
runpy.run_path(os.path.join(root, "bench.py"), run_name="__main__")
PY
