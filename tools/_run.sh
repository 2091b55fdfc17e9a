python -m pytest tests/test_gpu_wide.py -x -q 2>&1 | grep -E "^E  |Error|passed|failed" | head
python tools/quick_time_wide.py 32 8192 32 20 fp16
python tools/quick_time_wide.py 32 1024 32 20 fp16 | grep nll
python tools/quick_time_wide.py 32 1024 64 10 fp16 | grep nll
python tools/quick_time_wide.py 16 8192 32 20 fp16 | grep nll
