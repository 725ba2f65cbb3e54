#!/bin/bash
out=gpurun_out/r04g; mkdir -p $out
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "linear_parity or cfg2_full or batch_ragged or back_to_back or ranged" > $out/parity.txt 2>&1; echo "rc=$?"; tail -3 $out/parity.txt
for i in 1 2 3; do
  timeout -k 10 120 python scripts/bench_stft.py --reps 40 --gap-ms 1 --kernel 0 10 2>&1 | grep kernel= | tee -a $out/ab.txt
done
for i in 1 2 3; do
 for k in 0 10; do
  timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-single-track --no-cpu-baseline --no-skeleton --kernel $k > $out/b_${k}_$i.out 2> $out/b_${k}_$i.err
  echo "kernel $k: $(tail -1 $out/b_${k}_$i.out | python -c "import json,sys; j=json.loads(sys.stdin.read()); r=j['roofline']; print(r['avg_launch_ms'], r['frac'], r['launch_ms_min'], j['ms_per_step'])")" | tee -a $out/ab.txt
 done
done
