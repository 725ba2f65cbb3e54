cd "$GRAFT_REPO_ROOT"
for r in 1 2; do
for n in 16384 32768; do
echo "== n_fft $n: staging in lane order (new) | by column (previous)"
python3 scripts/bench_stft.py --nfft $n --reps 10 --gap-ms 1 --kernel 15 | grep median
THESIA_AMD_LIB=scripts/variants/libthesia_amd_subw_prev.so python3 scripts/bench_stft.py --nfft $n --reps 10 --gap-ms 1 --kernel 15 | grep median
done
echo "== 12000 / 3000 / 16384: new | previous"
python3 scripts/bench_stft.py --nfft 16384 --win 12000 --hop 3000 --reps 10 --gap-ms 1 --kernel 15 | grep median
THESIA_AMD_LIB=scripts/variants/libthesia_amd_subw_prev.so python3 scripts/bench_stft.py --nfft 16384 --win 12000 --hop 3000 --reps 10 --gap-ms 1 --kernel 15 | grep median
done
