# A/B: n_fft 32768 — stft_subwave_kernel (default) against stft_block_kernel (selector 14).  usage: gpurun -- bash scripts/ab_r05/subwave_r5.sh
cd "$GRAFT_REPO_ROOT"
rocm-smi --showserial 2>/dev/null | grep -i serial | tail -1
for r in 1 2; do
echo "== n_fft 32768 / hop 8192 (128 ch x 30 s): default | 14"
python3 scripts/bench_stft.py --nfft 32768 --reps 10 --gap-ms 1 --kernel 0 14 | grep median
echo "== 19200 / 4800 / 32768: default | 14"
python3 scripts/bench_stft.py --nfft 32768 --win 19200 --hop 4800 --reps 10 --gap-ms 1 --kernel 0 14 | grep median
done
