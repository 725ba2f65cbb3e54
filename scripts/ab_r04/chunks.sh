#!/bin/bash
out=gpurun_out/r04d; mkdir -p $out
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "packed_f32 or linear_parity or cfg2_full" > $out/parity.txt 2>&1; tail -2 $out/parity.txt
# chunk-size sweep on the bench workload (tuning selector: chunk << 16 | waves << 8 | 2), 1 ms gaps, then selector 9
K() { echo $(( ($1<<16) + (12<<8) + 2 )); }
python scripts/bench_stft.py --reps 40 --gap-ms 1 --kernel 0 $(K 24) $(K 30) $(K 40) $(K 59) $(K 60) $(K 118) 9 0 > $out/chunks.txt 2>&1
cat $out/chunks.txt | grep kernel=
# the full cfg5 (1024 tracks): default chunks vs 60
python scripts/bench_stft.py --reps 10 --gap-ms 1 --tracks 1024 --kernel 0 $(K 40) $(K 60) $(K 118) 0 > $out/chunks1024.txt 2>&1
cat $out/chunks1024.txt | grep kernel=
