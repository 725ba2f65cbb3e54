#!/bin/bash
# Development tool (round 2): chunk-schedule change — parity, per-wave times, A/B against the previous build, chunk sweep.
out=gpurun_out/sched.txt
: > $out
timeout 900 python -m pytest tests -x -q -m gpu > gpurun_out/sched_tests.log 2>&1; echo "pytest rc=$?" >> $out; tail -3 gpurun_out/sched_tests.log >> $out
THESIA_AMD_LIB=scripts/variants/libthesia_amd_wt.so timeout 300 python scripts/wave_times.py >> $out 2>&1
for i in 1 2 3; do
  for lib in scripts/variants/libthesia_amd_base.so thesia_amd/libthesia_amd.so; do
    echo -n "$(basename $lib): " >> $out
    THESIA_AMD_LIB=$lib timeout 300 python scripts/bench_stft.py --reps 30 --gap-ms 1 2>&1 | tail -1 >> $out
  done
done
ks=""
for c in 8 12 16 20 24 30; do ks="$ks $((2 + 12*256 + c*65536))"; done
for rep in 1 2; do
  timeout 300 python scripts/bench_stft.py --reps 30 --gap-ms 1 --kernel $ks 2>&1 | grep "^kernel" | sed 's/(stft_wave_kernel) n_fft=2048 win=2048 hop=512://' >> $out
done
timeout 300 python scripts/bench_stft.py --reps 30 --gap-ms 1 --nfft 4096 2>&1 | tail -1 >> $out
timeout 300 python scripts/bench_stft.py --reps 30 --gap-ms 1 --nfft 1024 2>&1 | tail -1 >> $out
timeout 300 python scripts/bench_stft.py --reps 30 --gap-ms 1 --nfft 512 2>&1 | tail -1 >> $out
timeout 300 python scripts/bench_stft.py --reps 30 --gap-ms 1 --win 1920 --hop 480 2>&1 | tail -1 >> $out
timeout 300 python scripts/bench_stft.py --reps 30 --tracks 1 --seconds 60 2>&1 | tail -1 >> $out
