#!/bin/bash
# Development tool (round 2): alternate several library builds on one box.  usage: r02_ab3.sh <out> <reps> lib... -- [bench_stft args]
out=$1; reps=$2; shift 2
libs=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do libs+=("$1"); shift; done
[ "$1" == "--" ] && shift
for i in $(seq $reps); do
  for lib in "${libs[@]}"; do
    echo -n "$(basename $lib): " >> $out
    THESIA_AMD_LIB=$lib timeout 300 python scripts/bench_stft.py --reps 30 --gap-ms 1 "$@" 2>&1 | tail -1 >> $out
  done
done
