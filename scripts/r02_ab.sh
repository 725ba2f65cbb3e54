#!/bin/bash
# Development tool (round 2): A/B the STFT launch of two library builds on one box, alternating.
# usage: scripts/r02_ab.sh <out> <libA> <libB> [bench_stft args...]
out=$1; a=$2; b=$3; shift 3
for i in 1 2 3; do
  for lib in $a $b; do
    echo -n "$(basename $lib): " >> $out
    THESIA_AMD_LIB=$lib timeout 300 python scripts/bench_stft.py --reps 30 --gap-ms 1 "$@" 2>&1 | tail -1 >> $out
  done
done
