# fused mel epilogue with (selector 5) and without (0) the grid-aligned register reuse of the linear plans
for i in 1 2; do
python3 scripts/bench_stft.py --reps 30 --gap-ms 1 --sr 48000 --win 1920 --hop 480 --mel 0 --kernel 0 5
python3 scripts/bench_stft.py --reps 30 --gap-ms 1 --sr 44100 --win 1764 --hop 441 --mel 0 --kernel 0 5
python3 scripts/bench_stft.py --reps 30 --gap-ms 1 --sr 16000 --nfft 1024 --win 640 --hop 160 --mel 0 --seconds 90 --kernel 0 5
done
