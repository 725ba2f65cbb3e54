# sweep schedule variants (kernel selector 11) against the default schedule (0): tails through one full-reload body (V2), + window from LDS (V6)
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "sweep" 2>&1 | tail -2
for lib in scripts/ab/libthesia_amd_swv6.so scripts/ab/libthesia_amd_swv2.so; do
echo "== $lib"
THESIA_AMD_LIB=$lib python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "sweep" 2>&1 | tail -1
for i in 1 2 3; do
THESIA_AMD_LIB=$lib python3 scripts/bench_stft.py --reps 30 --gap-ms 1 --kernel 0 11
done
done
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for k in 0 11; do
  THESIA_AMD_LIB=scripts/ab/libthesia_amd_swv6.so timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/swtrace_$k -- python3 scripts/bench_stft.py --reps 30 --gap-ms 1 --kernel $k > gpurun_out/swtrace_$k.log 2>&1
  f=$(find gpurun_out/swtrace_$k -name "*kernel_stats.csv" | head -1)
  echo "== V6 lib, kernel selector $k"
  python3 - "$f" <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if r["Name"].startswith(("th::","void th::")): print(f'  {r["Name"][:70]:70s} calls {r["Calls"]:>5s} avg {float(r["AverageNs"])/1e3:8.1f} us  min {float(r["MinNs"])/1e3:8.1f}  max {float(r["MaxNs"])/1e3:8.1f}')
PY
  find gpurun_out/swtrace_$k -type f -delete
done
