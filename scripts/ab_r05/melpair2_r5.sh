# A/B: frame-pair mel epilogue — default | one-frame epilogue (13) | pairs under the grid-aligned (phased / dynamic) frame loop (5)
# usage: gpurun -- bash scripts/ab_r05/melpair2_r5.sh
cd "$GRAFT_REPO_ROOT"
rocm-smi --showserial 2>/dev/null | grep -i serial | tail -1
for r in 1 2; do
echo "== app default mel (48 kHz, 1920 / 480, 347 mels): default | 13 | 5"
python3 scripts/bench_stft.py --sr 48000 --win 1920 --hop 480 --mel 0 --reps 20 --gap-ms 1 --kernel 0 13 5 | grep median
echo "== 44.1 kHz default mel (1764 / 441): default | 13 | 5"
python3 scripts/bench_stft.py --sr 44100 --win 1764 --hop 441 --mel 0 --reps 20 --gap-ms 1 --kernel 0 13 5 | grep median
echo "== cfg4 (44.1 kHz, 2048 / 512, mel-128, 32 tracks x 60 s): default | 13"
python3 scripts/bench_stft.py --sr 44100 --win 2048 --hop 512 --mel 128 --tracks 32 --seconds 60 --reps 20 --gap-ms 1 --kernel 0 13 | grep median
echo "== linear app default"
python3 scripts/bench_stft.py --sr 48000 --win 1920 --hop 480 --reps 20 --gap-ms 1 | grep median
done
