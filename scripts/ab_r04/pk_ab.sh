#!/bin/bash
# A/B on one box, alternating: the packed-f32 pipeline (product build) against the scalar one (scripts/ab/libthesia_amd_nopk.so)
out=gpurun_out/r04b; mkdir -p $out
python -m pytest tests -x -q -m gpu > $out/gputest.txt 2>&1; echo "pytest rc=$?" | tee -a $out/gputest.txt; tail -3 $out/gputest.txt
for i in 1 2 3; do
  for v in pk nopk; do
    if [ $v = nopk ]; then export THESIA_AMD_LIB=scripts/ab/libthesia_amd_nopk.so; else unset THESIA_AMD_LIB; fi
    echo "== $v gap1" >> $out/ab.txt; python scripts/bench_stft.py --reps 40 --gap-ms 1 >> $out/ab.txt 2>&1
    echo "== $v b2b" >> $out/ab.txt; python scripts/bench_stft.py --reps 40 >> $out/ab.txt 2>&1
  done
done
unset THESIA_AMD_LIB
cat $out/ab.txt
python bench.py --steps 20 --warmup 5 --no-single-track --no-cpu-baseline > $out/bench_pk.out 2> $out/bench_pk.err; tail -1 $out/bench_pk.out | cut -c1-1500
THESIA_AMD_LIB=scripts/ab/libthesia_amd_nopk.so python bench.py --steps 20 --warmup 5 --no-single-track --no-cpu-baseline > $out/bench_nopk.out 2> $out/bench_nopk.err; tail -1 $out/bench_nopk.out | cut -c1-1500
