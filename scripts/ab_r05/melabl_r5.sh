# A/B: app-default mel kernel (48 kHz, 1920 / 480, 347 mels) against the same kernel without its filterbank sums (TH_MELF_ABL=1),
# and the linear kernel of the same framing.  usage: gpurun -- bash scripts/ab_r05/melabl_r5.sh
cd "$GRAFT_REPO_ROOT"
bash scripts/box_id.sh 2>/dev/null | tail -2
for r in 1 2; do
echo "== default (mel)"; python3 scripts/bench_stft.py --sr 48000 --win 1920 --hop 480 --mel 0 --reps 20 | tail -1
echo "== no filterbank sums"; THESIA_AMD_LIB=scripts/variants/libthesia_amd_melabl1.so python3 scripts/bench_stft.py --sr 48000 --win 1920 --hop 480 --mel 0 --reps 20 | tail -1
echo "== mel, phased (selector 5)"; python3 scripts/bench_stft.py --sr 48000 --win 1920 --hop 480 --mel 0 --reps 20 --kernel 5 | tail -1
echo "== linear"; python3 scripts/bench_stft.py --sr 48000 --win 1920 --hop 480 --reps 20 | tail -1
echo "== linear, no phased mode (selector 4)"; python3 scripts/bench_stft.py --sr 48000 --win 1920 --hop 480 --reps 20 --kernel 4 | tail -1
done
