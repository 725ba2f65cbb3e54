# chunk length (frames per workgroup launch) of the block / planar / subwave plans: the default (total / (n_cu x 12), as for the wave kernels) against longer chunks
# (selector bits 16-23).  usage: gpurun -- bash scripts/ab_r05/block_chunks_r5.sh
cd "$GRAFT_REPO_ROOT"
rocm-smi --showserial 2>/dev/null | grep -i serial | tail -1
k() { echo $(( $1 + ($2 << 16) )); }
for r in 1 2; do
echo "== n_fft 8192: default | 60 | 90 | 120"; python3 scripts/bench_stft.py --nfft 8192 --reps 10 --gap-ms 1 --kernel 0 $(k 0 60) $(k 0 90) $(k 0 120) | grep median | cut -c1-110
echo "== n_fft 16384: default | 22 | 44 | 87"; python3 scripts/bench_stft.py --nfft 16384 --reps 10 --gap-ms 1 --kernel 0 $(k 0 22) $(k 0 44) $(k 0 87) | grep median | cut -c1-110
echo "== n_fft 32768: default | 22 | 43 | 86"; python3 scripts/bench_stft.py --nfft 32768 --reps 10 --gap-ms 1 --kernel 0 $(k 0 22) $(k 0 43) $(k 0 86) | grep median | cut -c1-110
echo "== n_fft 65536: default | 11 | 22 | 44"; python3 scripts/bench_stft.py --nfft 65536 --reps 6 --gap-ms 1 --kernel 0 $(k 0 11) $(k 0 22) $(k 0 44) | grep median | cut -c1-110
done
