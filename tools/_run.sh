for i in 1 2; do
for v in q8 q12 q16 q24; do NF_TOOL_LIB=noise_flow_amd/csrc/libnf_$v.so python tools/ab_pool.py 1024 2000 | tail -1; done
done
for v in q8 q16; do NF_TOOL_LIB=noise_flow_amd/csrc/libnf_$v.so python tools/quick_time.py 4096 50 | tail -2; done
