cd "$GRAFT_REPO_ROOT"
for r in 1 2; do
echo "== resident samples (default)"; python3 scripts/bench_stft.py --nfft 32768 --reps 10 --gap-ms 1 | grep median
echo "== every frame loaded in full"; THESIA_AMD_LIB=scripts/variants/libthesia_amd_subw_noreuse.so python3 scripts/bench_stft.py --nfft 32768 --reps 10 --gap-ms 1 | grep median
done
