#!/bin/bash
# Development tool: chunk-size sweep of the wave STFT kernel (bench workload), alternating, one box.
out=$1; shift
ks=""
for c in "$@"; do ks="$ks $((2 + 12*256 + c*65536))"; done
for rep in 1 2 3; do
  timeout 300 python scripts/bench_stft.py --reps 30 --gap-ms 1 --kernel $ks 2>&1 | grep "^kernel" | sed 's/(stft_wave_kernel) n_fft=2048 win=2048 hop=512://' >> $out
done
