#!/bin/bash
# Development tool: A/B of two library builds on config 4 (fused mel epilogue), alternating, on one box.
# usage: scripts/ab_cfg4.sh <libA> <libB> [rounds]
a=$1; b=$2; n=${3:-3}
for i in $(seq $n); do
  for lib in $a $b; do
    echo "== $lib"
    THESIA_AMD_LIB=$lib GAP_MS=1 python3 scripts/bench_cfg4.py 2>&1 | grep cfg4
    THESIA_AMD_LIB=$lib GAP_MS=1 SR=48000 WIN=1920 HOP=480 python3 scripts/bench_cfg4.py 2>&1 | grep "mel-347"
  done
done
