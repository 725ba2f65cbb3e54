#!/bin/bash
# rocprofv3 PMC passes over the STFT microbenchmark (run on the GPU box via gpurun).
# usage: scripts/pmc_stft.sh <outdir> [bench_stft args...]
set -u
out=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$out"
i=0
for ctrs in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU" \
            "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
            "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE SQ_INST_CYCLES_VMEM SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA"; do
  i=$((i+1))
  timeout 150 rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d "$out/pass$i" -- python3 ${TH_PMC_SCRIPT:-scripts/bench_stft.py} --reps 3 "$@" > "$out/pass$i.log" 2>&1
done
python3 - "$out" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/pass*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if not any(t in k for t in ("stft", "spec_to_img", "raster", "waveform", "mel_")):
            continue
        agg[k.split("(")[0][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(out + "/summary.txt", "w") as fo:
    for k, d in agg.items():
        fo.write(k + "\n")
        for c, v in sorted(d.items()):
            v = sorted(v)
            fo.write(f"  {c:28s} n={len(v):3d} median={v[len(v)//2]:.6g} max={v[-1]:.6g}\n")
print(open(out + "/summary.txt").read())
PY
# the raw per-dispatch tables are large (every torch kernel of the signal generator is in them): keep the summary only
[ -n "${TH_PMC_KEEP_RAW:-}" ] || rm -rf "$out"/pass*/
