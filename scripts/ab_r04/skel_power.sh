# package power and clocks while (a) the real kernel, (b) its memory skeleton with the stand-in work, (c) the same under the sweep schedule run back to back
for k in real skelA skelSweep; do
  case $k in
    real) bash scripts/power_probe.sh gpurun_out/pw_$k python3 scripts/bench_stft.py --reps 20000;;
    skelA) bash scripts/power_probe.sh gpurun_out/pw_$k scripts/ubench/stft_skeleton 0 9 0;;
    skelSweep) bash scripts/power_probe.sh gpurun_out/pw_$k scripts/ubench/stft_skeleton 0 9 1;;
  esac
  echo "== $k loop: $(grep -v amdgpu.ids gpurun_out/pw_$k.cmd | tail -1)"; grep -E "Power|sclk" gpurun_out/pw_$k | tail -6
done
