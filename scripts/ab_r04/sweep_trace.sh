# kernel-level durations of the default and the sweep schedule (rocprofv3 kernel trace of bench_stft.py)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for k in 0 11; do
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/swtrace_$k -- python3 scripts/bench_stft.py --reps 30 --gap-ms 1 --kernel $k > gpurun_out/swtrace_$k.log 2>&1
  f=$(find gpurun_out/swtrace_$k -name "*kernel_stats.csv" | head -1)
  echo "== kernel selector $k"; grep -v amdgpu.ids gpurun_out/swtrace_$k.log | tail -1
  python3 - "$f" <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if r["Name"].startswith(("th::","void th::")): print(f'  {r["Name"][:70]:70s} calls {r["Calls"]:>5s} avg {float(r["AverageNs"])/1e3:8.1f} us  min {float(r["MinNs"])/1e3:8.1f}  max {float(r["MaxNs"])/1e3:8.1f}')
PY
  find gpurun_out/swtrace_$k -type f -delete
done
