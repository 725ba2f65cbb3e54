# A/B: frame-pair mel epilogue at n_fft 1024 (default) against the one-frame epilogue (selector 13).  usage: gpurun -- bash scripts/ab_r05/melpair1024_r5.sh
cd "$GRAFT_REPO_ROOT"
rocm-smi --showserial 2>/dev/null | grep -i serial | tail -1
for r in 1 2; do
echo "== 16 kHz default (640 / 160 / 1024, mel default): default | 13"
python3 scripts/bench_stft.py --sr 16000 --nfft 1024 --win 640 --hop 160 --mel 0 --seconds 90 --reps 20 --gap-ms 1 --kernel 0 13 | grep median
echo "== 22.05 kHz default (882 / 220 / 1024): default | 13"
python3 scripts/bench_stft.py --sr 22050 --nfft 1024 --win 882 --hop 220 --mel 0 --seconds 60 --reps 20 --gap-ms 1 --kernel 0 13 | grep median
echo "== 48 kHz, 1024 / 256 / 1024, mel-128: default | 13"
python3 scripts/bench_stft.py --sr 48000 --nfft 1024 --mel 128 --reps 20 --gap-ms 1 --kernel 0 13 | grep median
done
