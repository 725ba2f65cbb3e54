#!/bin/bash
# package power and shader clock of the headline kernel's three builds INSIDE bench.py's step (STFT -> range ->
# fused image kernel, 128 tracks x 30 s), on one card, in one file: default | packed-f32 (selector 9) | sweep schedule (11).
# Needs an A/B build of the library:  VARIANT_DIR=../../scripts/ab VARIANT_SOURCES="kernels_stft.hip kernels_stft_long.hip api.hip" \
#   scripts/build_variant.sh ab -DTH_AB_VARIANTS=1        (scripts/ab/ travels to the GPU box)
# Run on the GPU box (gpurun): scripts/power_in_step.sh  ->  gpurun_out/power_in_step.txt
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
lib=scripts/ab/libthesia_amd_ab.so
out=gpurun_out/power_in_step
mkdir -p "$out"
{
  echo "# scripts/power_in_step.sh — card $(rocm-smi --showserial 2>/dev/null | grep -oE '[0-9]{9,}' | head -1); library $lib (-DTH_AB_VARIANTS=1)"
  echo "# per selector: bench.py --kernel K --steps 12000 (about 15 s of steps), rocm-smi sampled six times from second 6 on"
  for k in 0 9 11 0 9 11; do
    THESIA_AMD_LIB=$lib bash scripts/power_probe.sh "$out/pw_$k" python3 bench.py --kernel $k --steps 12000 --warmup 5 --no-cpu-baseline --no-single-track --no-skeleton --no-full-cfg5
    python3 - "$out/pw_$k" $k <<'PY'
import json, re, sys
p, k = sys.argv[1], sys.argv[2]
line = None
for l in reversed(open(p + ".cmd").read().splitlines()):
    if l.startswith("{") and '"metric"' in l:
        line = json.loads(l); break
txt = open(p).read()
pw = [float(x) for x in re.findall(r"Package Power \(W\): ([0-9.]+)", txt)]
ck = [int(x) for x in re.findall(r"sclk clock level: \d+: \((\d+)Mhz\)", txt)]
if line is None:
    print(f"selector {k}: no bench line"); sys.exit(0)
rf = line["roofline"]
ms, frames = line["ms_per_step"], line["config"]["frames_per_gpu"]
w = sum(pw) / max(1, len(pw))
print(f"selector {k:>2}: step {ms:.4f} ms, STFT launch {rf['avg_launch_ms']:.4f} ms (frac {rf['frac']:.4f}), image stage {rf['image_stage_in_step_ms']:.4f} ms | "
      f"package power {w:.0f} W (samples {[int(x) for x in pw]}), sclk {sorted(set(ck))} MHz | {w * ms * 1e-3 / frames * 1e6:.2f} uJ per frame for the whole step")
PY
    rm -f "$out/pw_$k" "$out/pw_$k.cmd" "$out/pw_$k.idle"
  done
} > gpurun_out/power_in_step.txt 2>&1
rmdir "$out" 2>/dev/null
cat gpurun_out/power_in_step.txt
