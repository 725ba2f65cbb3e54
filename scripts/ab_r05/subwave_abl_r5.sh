# ablation of stft_subwave_kernel (n_fft 32768 / hop 8192): TH_SUBW_ABL bits 1 no window loads, 2 no sample fetch, 4 no sub-transform,
# 8 no combining butterfly, 16 no split pass / rows.  usage: gpurun -- bash scripts/ab_r05/subwave_abl_r5.sh
cd "$GRAFT_REPO_ROOT"
rocm-smi --showserial 2>/dev/null | grep -i serial | tail -1
echo "== product"; python3 scripts/bench_stft.py --nfft 32768 --reps 10 --gap-ms 1 | grep median
for a in 1 2 3 4 8 16 28 31; do
echo "== abl $a"; THESIA_AMD_LIB=scripts/variants/libthesia_amd_subw_abl$a.so python3 scripts/bench_stft.py --nfft 32768 --reps 10 --gap-ms 1 | grep median
done
