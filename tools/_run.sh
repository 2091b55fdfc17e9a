mkdir -p gpurun_out/r2d
python -m pytest tests/test_gpu_wide.py -q 2>&1 | tail -4
for v in w3 w2 nodpp; do echo "== $v"; NF_TOOL_LIB=noise_flow_amd/csrc/libnf_$v.so python tools/quick_time_wide.py 32 8192 32 2>&1 | grep nll; done
echo "== default"; python tools/quick_time_wide.py 32 8192 32 | grep "nll\|sample"; python tools/quick_time_wide.py 32 1024 32 | grep nll;  python tools/quick_time_wide.py 32 1024 64 10 | grep nll
