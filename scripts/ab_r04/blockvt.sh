# block kernels (n_fft 16384 / 32768) with two virtual threads per thread (product) against round 3's shapes (vt1) and VT = 2 without the resident samples (vt2nr)
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "very_long or 16384 or block or wave_vs_generic or interior or calc_spec" 2>&1 | tail -2
for i in 1 2; do
for lib in "" scripts/ab/libthesia_amd_vt1.so scripts/ab/libthesia_amd_vt2nr.so; do
echo "== lib=${lib:-product}"
THESIA_AMD_LIB=$lib python3 scripts/bench_stft.py --reps 30 --gap-ms 1 --nfft 16384
THESIA_AMD_LIB=$lib python3 scripts/bench_stft.py --reps 20 --gap-ms 1 --nfft 32768
THESIA_AMD_LIB=$lib python3 scripts/bench_stft.py --reps 20 --gap-ms 1 --nfft 16384 --win 9600 --hop 2400
THESIA_AMD_LIB=$lib python3 scripts/bench_stft.py --reps 20 --gap-ms 1 --nfft 32768 --win 19200 --hop 4800
THESIA_AMD_LIB=$lib python3 scripts/bench_stft.py --reps 20 --gap-ms 1 --sr 48000 --nfft 16384 --mel 0
done
done
