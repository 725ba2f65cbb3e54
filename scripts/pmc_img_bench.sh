#!/bin/bash
# counters of spec_to_img_raster_kernel under bench.py's OWN workload (synthetic tracks through the STFT, not
# bench_img.py's uniformly random pixels), with the card's serial and the image stage's in-step time of this very box in one file.
# Run on the GPU box (gpurun): scripts/pmc_img_bench.sh  ->  gpurun_out/pmc_img_bench_<serial>.txt
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
serial=$(rocm-smi --showserial 2>/dev/null | grep -oE "[0-9]{9,}" | head -1)
serial=${serial:-unknown}
out=gpurun_out/pmc_img_bench_$serial
mkdir -p "$out"
lean="--steps 20 --warmup 5 --no-cpu-baseline --no-single-track --no-skeleton --no-full-cfg5"
python3 bench.py $lean > "$out/bench.json" 2> "$out/bench.err"
i=0
for ctrs in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU" \
            "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
            "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_sum TCC_EA0_RDREQ_sum" \
            "TCC_EA0_WRREQ_LEVEL_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_RDREQ_32B_sum" \
            "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCC_REQ_sum TCC_TAG_STALL_sum"; do
  i=$((i+1))
  timeout 240 rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d "$out/pass$i" -- python3 bench.py --steps 3 --warmup 1 --spin-up-steps 2 --no-cpu-baseline --no-single-track --no-skeleton --no-full-cfg5 > "$out/pass$i.log" 2>&1
done
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
line = None
for l in reversed(open(out + "/bench.json").read().splitlines()):
    if l.startswith("{") and '"metric"' in l:
        line = json.loads(l)
        break
with open(out + ".txt", "w") as fo:
    fo.write(f"# scripts/pmc_img_bench.sh: rocprofv3 --kernel-trace --pmc <group> -- python3 bench.py --steps 3 --warmup 1 --spin-up-steps 2 (bench.py's own workload)\n")
    fo.write(f"# card serial {serial}\n")
    if line:
        rf = line["roofline"]
        fo.write(f"# this box, bench.py --steps 20 --warmup 5 (no profiler): {line['value'] / 1e6:.1f} M frames/s, {line['ms_per_step']:.4f} ms per step, "
                 f"image_stage_in_step_ms {rf.get('image_stage_in_step_ms')}, STFT launch {rf.get('avg_launch_ms')} ms (frac {rf.get('frac')}), copy {rf.get('measured_copy_GBs')} GB/s\n")
    fo.write("# per-dispatch medians; SQ_* summed over waves, FETCH_SIZE / WRITE_SIZE in KB (FETCH_SIZE reports half of the streamed bytes on gfx950)\n")
    for k, d in agg.items():
        fo.write(k + "\n")
        for c, v in sorted(d.items()):
            v = sorted(v)
            fo.write(f"  {c:28s} n={len(v):3d} median={v[len(v)//2]:.6g} max={v[-1]:.6g}\n")
print(open(out + ".txt").read())
PY
rm -rf "$out"
