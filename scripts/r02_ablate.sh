#!/bin/bash
# Development tool (round 2): ablation matrix of the wave STFT kernel + the memory skeleton, one GPU call.
out=gpurun_out/r02_ablate.txt
mkdir -p gpurun_out
: > $out
for v in "" nostore smallwav nomem noex1 noex2 nodft nolog nocomp nocompmem; do
  lib=scripts/variants/libthesia_amd_$v.so; [ -z "$v" ] && lib=thesia_amd/libthesia_amd.so
  [ -f $lib ] || { echo "missing $lib" >> $out; continue; }
  echo "== variant '${v:-base}'" >> $out
  THESIA_AMD_LIB=$lib timeout 300 python scripts/bench_stft.py --reps 30 2>&1 | tail -1 >> $out
  THESIA_AMD_LIB=$lib timeout 300 python scripts/bench_stft.py --reps 30 --gap-ms 1 2>&1 | tail -1 >> $out
done
echo "== hop 1024 / 256 (base)" >> $out
timeout 300 python scripts/bench_stft.py --reps 30 --hop 1024 2>&1 | tail -1 >> $out
timeout 300 python scripts/bench_stft.py --reps 30 --hop 256 2>&1 | tail -1 >> $out
echo "== skeleton, back to back" >> $out
timeout 300 scripts/ubench/stft_skeleton 0 >> $out 2>&1
echo "== skeleton, 1 ms gaps" >> $out
timeout 300 scripts/ubench/stft_skeleton 1000 >> $out 2>&1
echo "== bench.py" >> $out
timeout 600 python bench.py >> $out 2>&1
