#!/bin/bash
out=gpurun_out/sched2.txt
: > $out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -x -q -m gpu > gpurun_out/sched_tests.log 2>&1; echo "pytest rc=$?" >> $out; tail -2 gpurun_out/sched_tests.log >> $out
THESIA_AMD_LIB=scripts/variants/libthesia_amd_wt.so timeout 300 python scripts/wave_times.py 2>&1 | head -26 >> $out
bash scripts/r02_ab3.sh $out 3 scripts/variants/libthesia_amd_base.so scripts/variants/libthesia_amd_noprio.so thesia_amd/libthesia_amd.so scripts/variants/libthesia_amd_pa4.so
ks=""
for c in 12 16 20 24 30; do ks="$ks $((2 + 12*256 + c*65536))"; done
for rep in 1 2; do
  timeout 300 python scripts/bench_stft.py --reps 30 --gap-ms 1 --kernel $ks 2>&1 | grep "^kernel" | sed 's/(stft_wave_kernel) n_fft=2048 win=2048 hop=512://' >> $out
done
