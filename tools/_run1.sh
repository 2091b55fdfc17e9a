mkdir -p gpurun_out/r06c
LIBS="base product" M=fp16,fp32 CFGS="64 1024 400;32 1024 2000;32 16384 100" bash tools/ab_lib.sh 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Host\|^Librccl" | tee gpurun_out/r06c/ab.log
