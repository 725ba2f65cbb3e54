#!/bin/bash
# Development tool: the multi-frame wave kernel (n_fft 512 / 1024) against the kernels it replaces, one box.
out=$1
for rep in 1 2; do
  # n_fft 1024 / hop 256: multi (auto) vs one-frame plan (selector 6) vs generic (1)
  timeout 300 python scripts/bench_stft.py --reps 20 --gap-ms 1 --nfft 1024 --kernel 0 6 2>&1 | grep "^kernel" >> $out
  timeout 300 python scripts/bench_stft.py --reps 20 --gap-ms 1 --nfft 512 --kernel 0 1 2>&1 | grep "^kernel" >> $out
  timeout 300 python scripts/bench_stft.py --reps 20 --gap-ms 1 --nfft 512 --win 320 --hop 80 --kernel 0 1 2>&1 | grep "^kernel" >> $out
  timeout 300 python scripts/bench_stft.py --reps 20 --gap-ms 1 --nfft 1024 --win 960 --hop 240 --kernel 0 6 2>&1 | grep "^kernel" >> $out
done
