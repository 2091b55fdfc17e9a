python -m pytest tests/test_gpu_batchstats.py tests/test_gpu_parity.py tests/test_gpu_wide.py -x -q 2>&1 | tail -3
python tools/time_batchstats.py
