#!/bin/bash
out=gpurun_out/r04c; mkdir -p $out
THESIA_AMD_LIB=scripts/ab/libthesia_amd_pknonop.so python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "linear_parity or cfg2 or wave_and_generic or baseline_sizes" > $out/parity_nonop.txt 2>&1; tail -2 $out/parity_nonop.txt
for i in 1 2 3; do
  for v in pk pknonop nopk; do
    if [ $v = pk ]; then unset THESIA_AMD_LIB; else export THESIA_AMD_LIB=scripts/ab/libthesia_amd_$v.so; fi
    python bench.py --steps 20 --warmup 5 --no-single-track --no-cpu-baseline --no-skeleton > $out/b_${v}_$i.out 2> $out/b_${v}_$i.err
    echo "$v $i $(tail -1 $out/b_${v}_$i.out | python -c "import json,sys; j=json.loads(sys.stdin.read()); r=j['roofline']; print(r['avg_launch_ms'], r['frac'], r['launch_ms_min'], j['ms_per_step'])")" | tee -a $out/ab.txt
  done
done
