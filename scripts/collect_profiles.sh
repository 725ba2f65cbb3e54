#!/bin/bash
# Run on the GPU box (gpurun): collects every measurement that profiles/ and the docs quote into gpurun_out/final/.
# (build the instrumented variant first if the phase profile is wanted: scripts/build_variant.sh prof -DTH_PHASE_PROF)
# usage: scripts/collect_profiles.sh        then locally: python scripts/make_profiles.py gpurun_out/final r01
# Since round 4 in parts (a gpurun call is at most 20 minutes): `collect_profiles.sh 1` = tests, bench lines, rocprofv3 trace, the
# PMC passes of the headline kernel (default / packed-f32 / sweep variants) and of the image kernels, power; `2` = the PMC
# passes of the other plans; `3` = the bench_stft / cfg tables and the micro-benchmarks.  No argument: everything.
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/final
part=${1:-all}
want() { [ "$part" = all ] || [ "$part" = "$1" ]; }
[ "$part" = all ] || [ "$part" = 1 ] && rm -rf "$out"
mkdir -p "$out"
if want 1; then
# which card, and whether the library that runs is the one that travelled with the push or a rebuild on this box
{ rocm-smi --showserial 2>/dev/null | grep -E "Serial"; } > "$out/box.txt"
python3 - > "$out/build_mode.txt" <<'PY'
import os, time
lib = "thesia_amd/libthesia_amd.so"
before = os.path.getmtime(lib) if os.path.exists(lib) else None
t0 = time.time()
import __graft_entry__ as ge
ge.build()
after = os.path.getmtime(lib)
print("build_mode: %s (__graft_entry__.build() took %.1f s; %s)" % (
    "reused the prebuilt library that travelled with the push" if before == after else "REBUILT on this box", time.time() - t0, lib))
PY
python3 -m pytest tests -x -q -m gpu -s 2>&1 | grep -E "passed|failed|error|moment-form" | tail -30 > "$out/gputest.txt"
echo "bench" >> "$out/progress.txt"; python3 bench.py --gpus 1 --steps 20 --warmup 5 > "$out/bench.json" 2> "$out/bench.err"
cp gpurun_out/bench_extras.json "$out/bench_extras.json" 2>/dev/null
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-single-track --no-skeleton > "$out/bench_under_rocprof.json" 2> "$out/trace.err"
find "$out/trace" -type f ! -name "*kernel_stats.csv" -delete
scripts/pmc_stft.sh "$out/pmc_stft" > "$out/pmc_stft.log" 2>&1
# (round 6: the packed-f32 pipeline and the sweep schedule — selectors 9 / 11 — are A/B builds only; their counters are in profiles/r05_stftpk_* / r05_stftsweep_*)
TH_PMC_SCRIPT=scripts/bench_img.py scripts/pmc_stft.sh "$out/pmc_img" > "$out/pmc_img.log" 2>&1
# the N > 1 launcher (round 5): `--gpus 2` on this one-GPU box must fail loudly, and the launcher -> torchrun -> rank -> RCCL route
# end to end at one rank (TH_BENCH_FORCE_LAUNCHER=1)
python3 bench.py --gpus 2 --steps 2 > "$out/gpus2.out" 2> "$out/gpus2.err"; echo "exit code $? ; stdout bytes $(wc -c < $out/gpus2.out)" >> "$out/gpus2.err"
TH_BENCH_FORCE_LAUNCHER=1 timeout -k 10 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-single-track --no-full-cfg5 > "$out/launcher.out" 2> "$out/launcher.err"; grep "^{" "$out/launcher.out" | tail -1 > "$out/bench_line_launcher.json"
# N > 1 control flow with real processes on this one card (REHEARSAL over gloo, every rank on GPU 0 — not a scaling figure)
for n in 2 4; do TH_BENCH_SHARE_GPU=1 timeout -k 10 400 python3 bench.py --gpus $n --steps 20 --warmup 5 --no-cpu-baseline > "$out/share$n.out" 2> "$out/share$n.err"; grep "^{" "$out/share$n.out" | tail -1 > "$out/bench_line_rehearsal_${n}_ranks_one_gpu.json"; done
# the RCCL path on one GPU (world size 1): the line bench.py prints with the process group up
TH_BENCH_FORCE_DIST=1 timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-single-track --no-full-cfg5 > "$out/force_dist.out" 2> "$out/force_dist.err"; grep "^{" "$out/force_dist.out" | tail -1 > "$out/bench_line_force_dist.json"
# package power / clocks next to the kernel loop and next to plain memory streams (scripts/power_probe.sh)
{
  for k in stft copy store; do
    if [ $k = stft ]; then bash scripts/power_probe.sh "$out/pw_$k" python3 scripts/bench_stft.py --reps 20000
    else bash scripts/power_probe.sh "$out/pw_$k" python3 scripts/power_loops.py $k 12; fi
    echo "== $k loop: $(tail -1 $out/pw_$k.cmd)"; grep -E "Power|sclk" "$out/pw_$k" | tail -4
  done
  echo "== idle"; grep -E "Max Graphics|Package Power" "$out/pw_stft.idle"
} > "$out/power.txt" 2>&1
fi
if want 2; then
scripts/pmc_stft.sh "$out/pmc_stft1024" --nfft 1024 > "$out/pmc_stft1024.log" 2>&1
scripts/pmc_stft.sh "$out/pmc_stft512_multi" --nfft 512 > "$out/pmc_stft512_multi.log" 2>&1
scripts/pmc_stft.sh "$out/pmc_stft4096" --nfft 4096 --seconds 60 > "$out/pmc_stft4096.log" 2>&1
scripts/pmc_stft.sh "$out/pmc_stftmel" --sr 44100 --tracks 32 --seconds 60 --mel 128 > "$out/pmc_stftmel.log" 2>&1
scripts/pmc_stft.sh "$out/pmc_stft4096dyn" --sr 96000 --nfft 4096 --win 3840 --hop 960 --seconds 30 > "$out/pmc_stft4096dyn.log" 2>&1
scripts/pmc_stft.sh "$out/pmc_stft512mel" --sr 8000 --nfft 512 --win 320 --hop 80 --mel 0 --seconds 180 > "$out/pmc_stft512mel.log" 2>&1   # (the 8 kHz Mel default: stft_wave_multi_kernel with the fused epilogue; rounds 3-5 filed it as "melrows")
# round 6: mel at n_fft 4096 — the moment-form epilogue (default) and round 5's two kernels (selector 12): mel_band_rows_kernel (96 kHz default, 404 mels) and mel_mfma_kernel (48 kHz, 695 mels)
scripts/pmc_stft.sh "$out/pmc_stft4096mel96" --sr 96000 --nfft 4096 --win 3840 --hop 960 --seconds 30 --mel 0 > "$out/pmc_stft4096mel96.log" 2>&1
scripts/pmc_stft.sh "$out/pmc_melbandrows96" --sr 96000 --nfft 4096 --win 3840 --hop 960 --seconds 30 --mel 0 --kernel 12 > "$out/pmc_melbandrows96.log" 2>&1
scripts/pmc_stft.sh "$out/pmc_stft4096mel48" --sr 48000 --nfft 4096 --mel 0 > "$out/pmc_stft4096mel48.log" 2>&1
scripts/pmc_stft.sh "$out/pmc_melmfma48" --sr 48000 --nfft 4096 --mel 0 --kernel 12 > "$out/pmc_melmfma48.log" 2>&1
scripts/pmc_stft.sh "$out/pmc_stftmel48" --sr 48000 --win 1920 --hop 480 --mel 0 > "$out/pmc_stftmel48.log" 2>&1
scripts/pmc_stft.sh "$out/pmc_stftmel48_one_frame" --sr 48000 --win 1920 --hop 480 --mel 0 --kernel 13 > "$out/pmc_stftmel48_one_frame.log" 2>&1  # round 5: the one-frame epilogue (the default takes frame pairs)
scripts/pmc_stft.sh "$out/pmc_subwave32768" --nfft 32768 > "$out/pmc_subwave32768.log" 2>&1   # round 5: stft_subwave_kernel
# round 6: the moment-form mel epilogue inside the workgroup-per-frame kernel (n_fft 16384 / 8192 Mel defaults) and the linear kernel beside it
scripts/pmc_stft.sh "$out/pmc_blockmel16384" --nfft 16384 --mel 0 > "$out/pmc_blockmel16384.log" 2>&1
scripts/pmc_stft.sh "$out/pmc_blockmel8192" --nfft 8192 --mel 0 > "$out/pmc_blockmel8192.log" 2>&1
scripts/pmc_stft.sh "$out/pmc_block16384" --nfft 16384 > "$out/pmc_block16384.log" 2>&1
# round 6, last day: the moment-form epilogue where no LDS table form exists (16 kHz under 85 ms: n_fft 2048, 771 mels; 16 kHz under 20 ms: n_fft 512, four frames per wave)
# and n_fft 4096 with the epilogue under hop 120 (48 kHz, 40 ms, t_overlap 16, f_overlap 2: the even-offset grid-aligned loop)
scripts/pmc_stft.sh "$out/pmc_melsmall2048" --sr 16000 --nfft 2048 --win 1360 --hop 340 --seconds 90 --mel 0 > "$out/pmc_melsmall2048.log" 2>&1
scripts/pmc_stft.sh "$out/pmc_melsmall512" --sr 16000 --nfft 512 --win 320 --hop 80 --seconds 90 --mel 0 > "$out/pmc_melsmall512.log" 2>&1
scripts/pmc_stft.sh "$out/pmc_mel4096hop120" --sr 48000 --nfft 4096 --win 1920 --hop 120 --mel 0 > "$out/pmc_mel4096hop120.log" 2>&1
fi
if want 3; then
{
  python3 scripts/bench_stft.py --reps 30 --kernel 0 1
  python3 scripts/bench_stft.py --reps 30 --nfft 1024
  python3 scripts/bench_stft.py --reps 30 --nfft 4096
  python3 scripts/bench_stft.py --reps 30 --win 1920 --hop 480 --kernel 0 4   # phased mode vs plain wave kernel
  python3 scripts/bench_stft.py --reps 30 --win 1764 --hop 441 --kernel 0 4   # dynamic mode vs plain wave kernel
  python3 scripts/bench_stft.py --reps 30 --win 1920 --hop 480 --gap-ms 1
  python3 scripts/bench_stft.py --reps 30 --win 1764 --hop 441 --gap-ms 1
  python3 scripts/bench_stft.py --reps 30 --gap-ms 1
  python3 scripts/bench_stft.py --reps 30 --hop 1024
} > "$out/bench_stft.txt" 2>&1
python3 scripts/bench_img.py --sustain 300 > "$out/bench_img.txt" 2>&1
python3 scripts/bench_cfg3.py > "$out/bench_cfg3.txt" 2>&1
{ python3 scripts/bench_cfg4.py; KERNEL=3 python3 scripts/bench_cfg4.py; SR=48000 WIN=1920 HOP=480 python3 scripts/bench_cfg4.py; } > "$out/bench_cfg4.txt" 2>&1
{
  python3 scripts/bench_stft.py --reps 30 --gap-ms 1 --nfft 512
  python3 scripts/bench_stft.py --reps 30 --gap-ms 1 --nfft 512 --win 320 --hop 80
  python3 scripts/bench_stft.py --reps 30 --gap-ms 1 --nfft 1024
  python3 scripts/bench_stft.py --reps 30 --gap-ms 1 --nfft 4096
  python3 scripts/bench_stft.py --reps 30 --gap-ms 1 --nfft 8192
  python3 scripts/bench_stft.py --reps 30 --gap-ms 1 --nfft 16384
  python3 scripts/bench_stft.py --reps 20 --gap-ms 1 --nfft 32768
  python3 scripts/bench_stft.py --reps 20 --gap-ms 1 --nfft 32768 --win 19200 --hop 4800
  python3 scripts/bench_stft.py --reps 20 --gap-ms 1 --nfft 16384 --win 12000 --hop 3000 --kernel 0 14
  python3 scripts/bench_stft.py --reps 20 --gap-ms 1 --nfft 16384 --kernel 0 15
  python3 scripts/bench_stft.py --reps 20 --gap-ms 1 --nfft 8192
  python3 scripts/bench_stft.py --reps 10 --gap-ms 1 --nfft 65536 --kernel 0 1   # round 5: planar block plan | generic kernel
  python3 scripts/bench_stft.py --reps 30 --gap-ms 1 --win 1920 --hop 240 --kernel 0 4
  python3 scripts/bench_stft.py --reps 30 --gap-ms 1 --win 1920 --hop 120 --kernel 0 4
  python3 scripts/bench_stft.py --reps 30 --gap-ms 1 --nfft 4096 --seconds 60
  python3 scripts/bench_stft.py --reps 30 --gap-ms 1 --nfft 8192 --win 3840 --hop 960 --sr 96000
  python3 scripts/bench_stft.py --reps 20 --gap-ms 1 --sr 8000 --nfft 512 --win 320 --hop 80 --mel 0 --seconds 180 --kernel 0 1
  python3 scripts/bench_stft.py --reps 20 --gap-ms 1 --sr 192000 --nfft 8192 --win 7680 --hop 1920 --mel 0 --seconds 8 --kernel 0 1
  python3 scripts/bench_stft.py --reps 30 --tracks 1 --seconds 60
  # round 3: grid-aligned reuse at n_fft 4096 (selector 4 = without), the banded-sum mel kernel (selector 7 = matrix cores)
  python3 scripts/bench_stft.py --reps 20 --gap-ms 1 --sr 96000 --nfft 4096 --win 3840 --hop 960 --seconds 30 --kernel 0 4
  python3 scripts/bench_stft.py --reps 20 --gap-ms 1 --sr 88200 --nfft 4096 --win 3528 --hop 882 --seconds 30 --kernel 0 4
  python3 scripts/bench_stft.py --reps 20 --gap-ms 1 --sr 96000 --nfft 4096 --win 3840 --hop 480 --seconds 30 --kernel 0 4
  python3 scripts/bench_stft.py --reps 20 --gap-ms 1 --sr 96000 --nfft 4096 --win 3840 --hop 960 --seconds 30 --mel 0 --kernel 0 4
  python3 scripts/bench_stft.py --reps 20 --gap-ms 1 --sr 8000 --nfft 512 --win 320 --hop 80 --mel 0 --seconds 180 --kernel 0 3 7
  python3 scripts/bench_stft.py --reps 20 --gap-ms 1 --sr 11025 --nfft 512 --win 441 --hop 110 --mel 0 --seconds 120 --kernel 0 3 7
  # the fused mel epilogue: banded sums (0) against pieces / gather (8)
  python3 scripts/bench_stft.py --reps 30 --gap-ms 1 --sr 48000 --win 1920 --hop 480 --mel 0 --kernel 0 8
  # round 5: the epilogue in frame pairs (0) against one frame at a time (13) and pairs under the grid-aligned frame loop (5)
  python3 scripts/bench_stft.py --reps 30 --gap-ms 1 --sr 48000 --win 1920 --hop 480 --mel 0 --kernel 0 13 5
  python3 scripts/bench_stft.py --reps 30 --gap-ms 1 --sr 44100 --tracks 32 --seconds 60 --mel 128 --kernel 0 13
  python3 scripts/bench_stft.py --reps 30 --gap-ms 1 --sr 44100 --win 1764 --hop 441 --mel 0 --tracks 32 --seconds 60 --kernel 0 8
  python3 scripts/bench_stft.py --reps 30 --gap-ms 1 --sr 44100 --tracks 32 --seconds 60 --mel 128 --kernel 0 8
  python3 scripts/bench_stft.py --reps 30 --gap-ms 1 --sr 16000 --nfft 1024 --win 640 --hop 160 --mel 0 --seconds 90 --kernel 0 8
  python3 scripts/bench_stft.py --reps 30 --gap-ms 1 --sr 22050 --nfft 1024 --win 882 --hop 220 --mel 0 --seconds 60 --kernel 0 8
  # round 6: mel at n_fft 4096 as the FFT kernel's epilogue in the moment form (0, default) against round 5's two kernels (12)
  python3 scripts/bench_stft.py --reps 20 --gap-ms 1 --sr 96000 --nfft 4096 --win 3840 --hop 960 --seconds 30 --mel 0 --kernel 0 12
  python3 scripts/bench_stft.py --reps 20 --gap-ms 1 --sr 88200 --nfft 4096 --win 3528 --hop 882 --seconds 30 --mel 0 --kernel 0 12
  python3 scripts/bench_stft.py --reps 20 --gap-ms 1 --sr 96000 --nfft 4096 --win 3840 --hop 480 --seconds 30 --mel 0 --kernel 0 12
  python3 scripts/bench_stft.py --reps 20 --gap-ms 1 --sr 48000 --nfft 4096 --mel 0 --kernel 0 12
  python3 scripts/bench_stft.py --reps 20 --gap-ms 1 --sr 48000 --nfft 4096 --hop 2048 --mel 0 --kernel 0 12
  # round 6: mel inside the workgroup-per-frame kernel (0, default) against the two kernels (12), 64 ch x 30 s
  python3 scripts/bench_stft.py --reps 20 --gap-ms 1 --tracks 64 --nfft 16384 --mel 0 --kernel 0 12
  python3 scripts/bench_stft.py --reps 20 --gap-ms 1 --tracks 64 --nfft 8192 --mel 0 --kernel 0 12
  python3 scripts/bench_stft.py --reps 20 --gap-ms 1 --tracks 64 --nfft 16384
  python3 scripts/bench_stft.py --reps 20 --gap-ms 1 --tracks 64 --nfft 8192
  # round 4: the Mel default of long windows (more than 512 mels) on the two-kernel path against the generic kernel
  python3 scripts/bench_stft.py --reps 20 --gap-ms 1 --sr 48000 --nfft 4096 --mel 0 --kernel 0 1
  python3 scripts/bench_stft.py --reps 20 --gap-ms 1 --sr 48000 --nfft 8192 --mel 0 --kernel 0 1
  python3 scripts/bench_stft.py --reps 20 --gap-ms 1 --sr 48000 --nfft 16384 --mel 0 --kernel 0 1
  python3 scripts/bench_stft.py --reps 10 --gap-ms 1 --sr 48000 --nfft 32768 --mel 0 --kernel 0 1
} >> "$out/bench_stft.txt" 2>&1
if [ -f scripts/variants/libthesia_amd_wt.so ]; then
  THESIA_AMD_LIB=scripts/variants/libthesia_amd_wt.so python3 scripts/wave_times.py > "$out/wave_times.txt" 2>&1
fi
[ -x scripts/ubench/mom_probe ] && { for a in "96000 4096 0" "88200 4096 0" "48000 4096 0"; do timeout 60 scripts/ubench/mom_probe $a 64; done; } > "$out/ubench_mom_probe.txt" 2>&1
timeout -k 10 200 python3 scripts/bluestein_probe.py 2>/dev/null | grep -v amdgpu.ids > "$out/bluestein_probe.txt"
timeout -k 10 200 python3 scripts/fuzz_mel_moments.py 6 90 2>/dev/null | grep -v "^refused\|amdgpu.ids" > "$out/fuzz_mel_moments.txt"
for u in lds_rate valu_rate valu_bank copy_rate row_stores stream_shapes fused_img_shapes; do
  [ -x scripts/ubench/$u ] && timeout 120 scripts/ubench/$u > "$out/ubench_$u.txt" 2>&1
done
[ -x scripts/ubench/stft_skeleton ] && timeout 120 scripts/ubench/stft_skeleton 1000 8 > "$out/ubench_stft_skeleton_sweep.txt" 2>&1
if [ -f scripts/variants/libthesia_amd_prof.so ]; then
  THESIA_AMD_LIB=scripts/variants/libthesia_amd_prof.so python3 scripts/phase_prof.py > "$out/phase_prof.txt" 2>&1
fi
fi
ls -la "$out"
