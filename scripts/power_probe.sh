#!/bin/bash
# Development tool: power / clocks of the GPU while a kernel loop runs.  usage: power_probe.sh <out> <cmd...>
out=$1; shift
rocm-smi --showpower --showclocks --showmaxpower --showperflevel > $out.idle 2>&1
"$@" > $out.cmd 2>&1 &
pid=$!
sleep 6
for i in 1 2 3 4 5 6; do
  rocm-smi --showpower --showclocks --showtemp 2>&1 | grep -E "Power|sclk|mclk|fclk|Temperature \(Sensor (junction|edge)" >> $out
  echo "--" >> $out
  sleep 1
done
wait $pid
