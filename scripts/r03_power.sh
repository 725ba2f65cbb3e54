#!/bin/bash
# Development tool (round 3): package power next to (a) the wave STFT kernel, (b) its build without global memory traffic,
# (c) the memory skeletons of scripts/ubench/stream_shapes.hip (row streams A, sweeps B).
out=gpurun_out/r03_power.txt
: > $out
probe() {  # label, command...
  label=$1; shift
  echo "== $label" >> $out
  rm -f /tmp/pp /tmp/pp.cmd /tmp/pp.idle
  scripts/power_probe.sh /tmp/pp "$@"
  tail -2 /tmp/pp.cmd >> $out
  grep -E "Power|sclk" /tmp/pp | sort | uniq -c | sort -rn | head -6 >> $out
}
probe "stft kernel" python scripts/bench_stft.py --reps 20000
THESIA_AMD_LIB=scripts/ab/libthesia_amd_nomem.so probe "stft kernel, no global memory traffic" python scripts/bench_stft.py --reps 20000
THESIA_AMD_LIB=scripts/ab/libthesia_amd_nostore.so probe "stft kernel, no stores" python scripts/bench_stft.py --reps 20000
for v in A B12x4 B8x4 B16x1; do probe "skeleton $v" scripts/ubench/stream_shapes loop $v 14; done
cat $out
