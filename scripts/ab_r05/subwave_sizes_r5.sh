# A/B: stft_subwave_kernel (selector 15) against stft_block_kernel (14) at n_fft 8192 / 16384 / 32768.  usage: gpurun -- bash scripts/ab_r05/subwave_sizes_r5.sh
cd "$GRAFT_REPO_ROOT"
rocm-smi --showserial 2>/dev/null | grep -i serial | tail -1
for r in 1 2; do
for n in 8192 16384 32768; do
echo "== n_fft $n, hop n_fft / 4: subwave (15) | block (14)"
python3 scripts/bench_stft.py --nfft $n --reps 10 --gap-ms 1 --kernel 15 14 | grep median
done
echo "== 96 kHz, 3840 / 960 / 8192: 15 | 14"
python3 scripts/bench_stft.py --nfft 8192 --win 3840 --hop 960 --sr 96000 --reps 10 --gap-ms 1 --kernel 15 14 | grep median
echo "== 48 kHz, 12000 / 3000 / 16384: 15 | 14"
python3 scripts/bench_stft.py --nfft 16384 --win 12000 --hop 3000 --reps 10 --gap-ms 1 --kernel 15 14 | grep median
done
