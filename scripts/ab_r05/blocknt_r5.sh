# A/B: non-temporal row stores in the block kernels (n_fft 8192 / 16384 / 65536, and 32768 under selector 14).  usage: gpurun -- bash scripts/ab_r05/blocknt_r5.sh
cd "$GRAFT_REPO_ROOT"
rocm-smi --showserial 2>/dev/null | grep -i serial | tail -1
for r in 1 2; do
for n in 8192 16384 65536; do
echo "== n_fft $n: nt (default) | plain"
python3 scripts/bench_stft.py --nfft $n --reps 10 --gap-ms 1 | grep median
THESIA_AMD_LIB=scripts/variants/libthesia_amd_blocknt0.so python3 scripts/bench_stft.py --nfft $n --reps 10 --gap-ms 1 | grep median
done
echo "== 96 kHz, 3840 / 960 / 8192: nt | plain"
python3 scripts/bench_stft.py --nfft 8192 --win 3840 --hop 960 --sr 96000 --reps 10 --gap-ms 1 | grep median
THESIA_AMD_LIB=scripts/variants/libthesia_amd_blocknt0.so python3 scripts/bench_stft.py --nfft 8192 --win 3840 --hop 960 --sr 96000 --reps 10 --gap-ms 1 | grep median
done
