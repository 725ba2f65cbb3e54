#!/bin/bash
out=gpurun_out/r04m; mkdir -p $out
{
  for k in "stft4096:--nfft 4096 --seconds 60" "stftmel48:--sr 48000 --win 1920 --hop 480 --mel 0" "stft1024:--nfft 1024" "stft8192:--nfft 8192"; do
    name=${k%%:*}; args=${k#*:}
    bash scripts/power_probe.sh "$out/pw_$name" python3 scripts/bench_stft.py --reps 12000 $args
    echo "== $name loop: $(tail -1 $out/pw_$name.cmd)"; grep -E "Power|sclk" "$out/pw_$name" | tail -4
  done
} > $out/power_more.txt 2>&1
cat $out/power_more.txt
