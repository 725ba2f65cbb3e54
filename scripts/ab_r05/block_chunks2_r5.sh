cd "$GRAFT_REPO_ROOT"
for r in 1 2; do
for n in 8192 16384 32768; do python3 scripts/bench_stft.py --nfft $n --reps 10 --gap-ms 1 | grep median | cut -c1-110; done
python3 scripts/bench_stft.py --nfft 8192 --win 3840 --hop 960 --sr 96000 --reps 10 --gap-ms 1 | grep median | cut -c1-110
python3 scripts/bench_stft.py --nfft 16384 --win 12000 --hop 3000 --reps 10 --gap-ms 1 | grep median | cut -c1-110
python3 scripts/bench_stft.py --nfft 16384 --tracks 1 --seconds 60 --reps 10 --gap-ms 1 | grep median | cut -c1-110
python3 scripts/bench_stft.py --nfft 32768 --tracks 2 --seconds 120 --reps 10 --gap-ms 1 | grep median | cut -c1-110
done
