# A/B: banded mel table in the paired layout (ds_read_b64 amplitudes, ds_read_b128 weights; product) against the plain layout (variant)
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "mel or track_manager" 2>&1 | tail -2
for i in 1 2 3; do
for lib in "" scripts/ab/libthesia_amd_plain.so; do
echo "== lib=${lib:-product}"
THESIA_AMD_LIB=$lib python3 scripts/bench_stft.py --reps 30 --gap-ms 1 --sr 48000 --win 1920 --hop 480 --mel 0
THESIA_AMD_LIB=$lib python3 scripts/bench_stft.py --reps 30 --gap-ms 1 --sr 44100 --tracks 32 --seconds 60 --mel 128
THESIA_AMD_LIB=$lib python3 scripts/bench_stft.py --reps 30 --gap-ms 1 --sr 44100 --win 1764 --hop 441 --mel 0
THESIA_AMD_LIB=$lib python3 scripts/bench_stft.py --reps 30 --gap-ms 1 --sr 48000 --mel 0
THESIA_AMD_LIB=$lib python3 scripts/bench_stft.py --reps 30 --gap-ms 1 --sr 16000 --nfft 1024 --win 640 --hop 160 --mel 0 --seconds 90
THESIA_AMD_LIB=$lib python3 scripts/bench_stft.py --reps 30 --gap-ms 1 --sr 22050 --nfft 1024 --win 882 --hop 221 --mel 0 --seconds 60
done
done
