python -m pytest tests -m gpu -x -q 2>&1 | tail -3
for i in 1 2; do
NF_TOOL_LIB=noise_flow_amd/csrc/libnf_base.so python tools/ab_pool.py 1024 2000 | tail -1
python tools/ab_pool.py 1024 2000 | tail -1
done
python tools/ab_pool.py 16384 200 | tail -1
NF_TOOL_LIB=noise_flow_amd/csrc/libnf_base.so python tools/ab_pool.py 16384 200 | tail -1
python tools/quick_time.py 4096 50
