R=$PWD
for rep in 1 2 3; do
  for lib in base new; do
    L=""; [ $lib = base ] && L=$R/build/variants/lib_base.so
    for cfg in "64 1024 400" "32 1024 2000"; do
      echo "$lib $cfg: $(NF_TOOL_LIB=$L MODES=${M:-fp16,fp32} python tools/quick_time_fp16.py $cfg | grep nll | tr '\n' ' ')"
    done
  done
done
