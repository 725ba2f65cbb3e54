#!/bin/bash
# kernel timeline of the last single-track step (run on the GPU box)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/trace_single; rm -rf $out
timeout 200 rocprofv3 --kernel-trace --output-format csv -d $out -- python3 scripts/trace_single.py > $out.log 2>&1
python3 - $(ls $out/*/*kernel_trace.csv | head -1) <<'PY'
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
rows = [r for r in rows if r["Kernel_Name"].startswith(("th::", "void th::"))]
# last step = kernels from the last STFT launch on
idx = max(i for i, r in enumerate(rows) if "stft_wave_kernel" in r["Kernel_Name"])
t0 = int(rows[idx]["Start_Timestamp"])
for r in rows[idx:]:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    print(f"{s/1e3:8.1f} us  +{(e-s)/1e3:7.1f} us  {r['Kernel_Name'][:60]}  grid {r['Grid_Size_X']} wg {r['Workgroup_Size_X']}")
PY
