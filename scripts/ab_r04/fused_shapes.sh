#!/bin/bash
out=gpurun_out/r04k; mkdir -p $out
for i in 1 2; do
for v in default fused_32_256 fused_32_512 fused_64_512 fused_128_1024; do
  if [ $v = default ]; then unset THESIA_AMD_LIB; else export THESIA_AMD_LIB=scripts/ab/libthesia_amd_$v.so; fi
  echo "== $v" | tee -a $out/ab.txt
  timeout -k 10 120 python scripts/bench_img.py 2>&1 | grep -E "fused|two kernels" | tee -a $out/ab.txt
done
done
