# n_fft 4096 grid-aligned mode without the odd window table where the frame offsets are always even (eight waves; product) against seven waves + two tables (noeven)
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "4096 or dynamic or phased or framing or mel or calc_spec or fuzz" 2>&1 | tail -2
for i in 1 2 3; do
for lib in "" scripts/ab/libthesia_amd_noeven.so; do
echo "== lib=${lib:-product}"
THESIA_AMD_LIB=$lib python3 scripts/bench_stft.py --reps 30 --gap-ms 1 --sr 96000 --nfft 4096 --win 3840 --hop 960 --seconds 30
THESIA_AMD_LIB=$lib python3 scripts/bench_stft.py --reps 30 --gap-ms 1 --sr 88200 --nfft 4096 --win 3528 --hop 882 --seconds 30
THESIA_AMD_LIB=$lib python3 scripts/bench_stft.py --reps 30 --gap-ms 1 --sr 96000 --nfft 4096 --win 3840 --hop 960 --mel 0 --seconds 30
done
done
