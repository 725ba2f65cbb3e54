#!/bin/bash
# Development tool: waves-per-CU / resident-table sweep of the wave STFT kernel (one box).
out=$1; shift
for rep in 1 2; do
for lib in "$@"; do
  echo "== $(basename $lib)" >> $out
  THESIA_AMD_LIB=$lib timeout 300 python scripts/bench_stft.py --reps 30 --gap-ms 1 --kernel 2050 2562 3074 3586 4098 2>&1 | grep "^kernel" | sed 's/(stft_wave_kernel) n_fft=2048 win=2048 hop=512://' >> $out
done
done
