#!/bin/bash
# is the n_fft 4096 kernel (cfg3, BASELINE's named config) at the power cap like the headline kernel?
# Package power and shader clock while each kernel runs back to back for ~15 s (rocm-smi, six samples from second 6 on):
# n_fft 4096 / hop 1024 (cfg3: 128 ch x 60 s), n_fft 2048 / hop 512 (the headline kernel, same audio), and the copy-like image kernel's
# neighbour for scale: n_fft 4096 with 30 ms pauses between launches (not power-limited).
# Run on the GPU box: scripts/power_cfg3.sh -> gpurun_out/power_cfg3.txt
set -u
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/power_cfg3
mkdir -p "$out"
{
  echo "# scripts/power_cfg3.sh — card $(rocm-smi --showserial 2>/dev/null | grep -oE '[0-9]{9,}' | head -1)"
  for cfg in "4096 14000 0" "2048 14000 0" "4096 400 30"; do
    set -- $cfg
    bash scripts/power_probe.sh "$out/pw" python3 scripts/bench_stft.py --nfft $1 --tracks 128 --seconds 60 --reps $2 --gap-ms $3
    echo "== n_fft $1, $2 launches, $3 ms between launches: $(grep 'kernel=' $out/pw.cmd | cut -c1-150)"
    grep -E "Package Power|sclk" "$out/pw" | sed 's/^GPU\[0\]\s*: //' | paste -sd' ' | sed 's/Current Socket Graphics Package Power (W): /W=/g; s/sclk clock level: [0-9]: //g'
    rm -f "$out/pw" "$out/pw.cmd" "$out/pw.idle"
  done
} > gpurun_out/power_cfg3.txt 2>&1
rmdir "$out" 2>/dev/null
cat gpurun_out/power_cfg3.txt
