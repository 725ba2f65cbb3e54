#!/bin/bash
# rocprofv3 PMC passes with caller-given counter groups over the STFT microbenchmark (run on the GPU box via gpurun).
# usage: scripts/pmc_groups.sh <outdir> "<group 1 counters>" ["<group 2 counters>" ...]     env TH_PMC_ARGS: bench_stft args
set -u
out=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$out"
i=0
for ctrs in "$@"; do
  i=$((i+1))
  timeout 150 rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d "$out/pass$i" -- python3 scripts/bench_stft.py --reps 3 ${TH_PMC_ARGS:-} > "$out/pass$i.log" 2>&1
done
python3 - "$out" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/pass*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "stft" not in k:
            continue
        agg[k.split("(")[0][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(out + "/summary.txt", "w") as fo:
    for k, d in agg.items():
        fo.write(k + "\n")
        for c, v in sorted(d.items()):
            v = sorted(v)
            fo.write(f"  {c:32s} n={len(v):3d} median={v[len(v)//2]:.6g} max={v[-1]:.6g}\n")
print(open(out + "/summary.txt").read())
PY
rm -rf "$out"/pass*/
