for i in 1 2 3; do
for lib in "" scripts/ab/libthesia_amd_rowspaired.so; do
echo "== lib=${lib:-product}"
THESIA_AMD_LIB=$lib python3 scripts/bench_stft.py --reps 30 --gap-ms 1 --sr 96000 --nfft 4096 --win 3840 --hop 960 --mel 0 --seconds 30
THESIA_AMD_LIB=$lib python3 scripts/bench_stft.py --reps 30 --gap-ms 1 --sr 88200 --nfft 4096 --win 3528 --hop 882 --mel 0 --seconds 30
done
done
THESIA_AMD_LIB=scripts/ab/libthesia_amd_rowspaired.so python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "mel" 2>&1 | tail -2
