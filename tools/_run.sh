python -m pytest tests/test_gpu_batchstats.py tests/test_gpu_scripts.py -x -q 2>&1 | grep -E "^E  |Error|passed|failed" | head
python tools/time_batchstats.py
