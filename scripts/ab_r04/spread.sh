# A/B: banded mel tables with the first bins spread over the LDS banks (product) against the unshifted tables (variant)
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "mel or track_manager" 2>&1 | tail -2
for i in 1 2 3; do
for lib in "" scripts/ab/libthesia_amd_nospread.so; do
echo "== lib=${lib:-product}"
THESIA_AMD_LIB=$lib python3 scripts/bench_stft.py --reps 30 --gap-ms 1 --sr 48000 --win 1920 --hop 480 --mel 0
THESIA_AMD_LIB=$lib python3 scripts/bench_stft.py --reps 30 --gap-ms 1 --sr 44100 --tracks 32 --seconds 60 --mel 128
THESIA_AMD_LIB=$lib python3 scripts/bench_stft.py --reps 30 --gap-ms 1 --sr 44100 --win 1764 --hop 441 --mel 0
THESIA_AMD_LIB=$lib python3 scripts/bench_stft.py --reps 30 --gap-ms 1 --sr 16000 --nfft 1024 --win 640 --hop 160 --mel 0 --seconds 90
THESIA_AMD_LIB=$lib python3 scripts/bench_stft.py --reps 30 --gap-ms 1 --sr 96000 --nfft 4096 --win 3840 --hop 960 --mel 0 --seconds 30
THESIA_AMD_LIB=$lib python3 scripts/bench_stft.py --reps 30 --gap-ms 1 --sr 8000 --nfft 512 --win 320 --hop 80 --mel 0 --seconds 180
done
done
