# A/B: running min / max of the linear wave kernels as v_min3_f32 / v_max3_f32 over two bins (default) against one v_min / v_max per bin (-DTH_MINMAX3=0).
# usage: gpurun -- bash scripts/ab_r05/minmax3_r5.sh     (bench.py lines alternate on one card: the STFT kernel inside the whole step)
cd "$GRAFT_REPO_ROOT"
rocm-smi --showserial 2>/dev/null | grep -i serial | tail -1
for r in 1 2 3; do
for v in default mm3off; do
if [ $v = default ]; then unset THESIA_AMD_LIB; else export THESIA_AMD_LIB=scripts/variants/libthesia_amd_$v.so; fi
python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-single-track --no-full-cfg5 --no-skeleton 2>/dev/null | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$v', 'value %.1f M step %.4f stft %.4f frac %.4f cfg3 %.4f app_default_linear %.4f' % (d['value']/1e6, d['ms_per_step'], r['avg_launch_ms'], r['frac'], r.get('cfg3_ms',0), r.get('app_default_linear_ms',0)))"
done
done
