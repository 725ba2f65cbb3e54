#!/bin/bash
# Round 6, last day: the image stage's two speeds (0.59-0.62 | 0.69-0.71 ms in bench.py's step) turned up on ONE card an hour apart (serial
# 692531011155: 0.698, then 0.597 ms) — not a property of the card.  Address translation is the suspect the earlier counter groups did not
# cover: TLB (UTCL1 / UTCL2) counters of spec_to_img_raster_kernel under bench.py's own workload, with this session's in-step time.
# Run on the GPU box (gpurun): scripts/pmc_img_tlb.sh  ->  gpurun_out/pmc_img_tlb_<serial>_<hhmmss>.txt
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
serial=$(rocm-smi --showserial 2>/dev/null | grep -oE "[0-9]{9,}" | head -1)
serial=${serial:-unknown}
out=gpurun_out/pmc_img_tlb_${serial}_$(date +%H%M%S)
mkdir -p "$out"
lean="--steps 20 --warmup 5 --no-cpu-baseline --no-single-track --no-skeleton --no-full-cfg5"
python3 bench.py $lean > "$out/bench.json" 2> "$out/bench.err"
i=0
for ctrs in "TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_PERMISSION_MISS_sum" \
            "GRBM_UTCL2_BUSY GRBM_GUI_ACTIVE" "TCP_UTCL1_SERIALIZATION_STALL TCP_UTCL1_TRANSLATION_MISS_UNDER_MISS TCP_UTCL1_STALL_INFLIGHT_MAX"; do
  i=$((i+1))
  timeout 240 rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d "$out/pass$i" -- python3 bench.py --steps 3 --warmup 1 --spin-up-steps 2 --no-cpu-baseline --no-single-track --no-skeleton --no-full-cfg5 > "$out/pass$i.log" 2>&1
done
python3 bench.py $lean > "$out/bench2.json" 2> "$out/bench2.err"
python3 - "$out" "$serial" <<'PY'
import csv, glob, json, sys, collections
out, serial = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/pass*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "spec_to_img_raster" not in k and "stft_wave_kernel" not in k:
            continue
        agg[k.split("(")[0][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
def line_of(p):
    for l in reversed(open(p).read().splitlines()):
        if l.startswith("{") and '"metric"' in l:
            return json.loads(l)
with open(out + ".txt", "w") as fo:
    fo.write("# scripts/pmc_img_tlb.sh: rocprofv3 --kernel-trace --pmc <group> -- python3 bench.py --steps 3 --warmup 1 --spin-up-steps 2 (bench.py's own workload)\n")
    fo.write(f"# card serial {serial}\n")
    for tag, p in (("before the counter passes", out + "/bench.json"), ("after them", out + "/bench2.json")):
        line = line_of(p)
        if line:
            rf = line["roofline"]
            fo.write(f"# this session, bench.py --steps 20 --warmup 5, {tag}: {line['value'] / 1e6:.1f} M frames/s, {line['ms_per_step']:.4f} ms per step, "
                     f"image_stage_in_step_ms {rf.get('image_stage_in_step_ms')}, STFT launch {rf.get('avg_launch_ms')} ms, copy {rf.get('measured_copy_GBs')} GB/s\n")
    for k, d in agg.items():
        fo.write(k + "\n")
        for c, v in sorted(d.items()):
            v = sorted(v)
            fo.write(f"  {c:40s} n={len(v):3d} median={v[len(v)//2]:.6g} max={v[-1]:.6g}\n")
print(open(out + ".txt").read())
PY
rm -rf "$out"
