# A/B: n_fft 65536 — stft_subwave2_kernel (default) against the planar block kernel (selector 14).  usage: gpurun -- bash scripts/ab_r05/subwave2_r5.sh
cd "$GRAFT_REPO_ROOT"
rocm-smi --showserial 2>/dev/null | grep -i serial | tail -1
for r in 1 2; do
echo "== n_fft 65536 / hop 16384: default | 14"; python3 scripts/bench_stft.py --nfft 65536 --reps 6 --gap-ms 1 --kernel 0 14 | grep median | cut -c1-130
echo "== 96 kHz, 38400 / 9600 / 65536: default | 14"; python3 scripts/bench_stft.py --sr 96000 --nfft 65536 --win 38400 --hop 9600 --reps 6 --gap-ms 1 --kernel 0 14 | grep median | cut -c1-130
done
