# Round 6, GPU call F: non-temporal output stores in the GEMM epilogues (gemm_bench A/B/A; engine two-build A/B), default bench line with the new probe
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp OMP_WAIT_POLICY=passive
OUT=gpurun_out/r06_f.txt; : > $OUT
echo "== gemm_bench store epilogue: plain / nt stores / plain, 40 launches per line" | tee -a $OUT
for B in gemm_bench gemm_bench_nt2 gemm_bench gemm_bench_nt2; do echo "-- $B" | tee -a $OUT; SS_GEMM_REPS=40 ./tools/$B.bin 2>&1 | grep -E ' (store|gelu) ' | grep -E '^(FC1|QK|crossKV|FC1x4|QKx4|Ox4) ' | tee -a $OUT; done
echo "== engine: cross-K/V cache stores nt (nt1), all 16-bit epilogue stores nt (nt3), alternating with the plain build" | tee -a $OUT
for rep in 1 2; do for which in plain nt1 nt3; do
  if [ $which = plain ]; then unset SS_LIB_PATH; else export SS_LIB_PATH=$PWD/gpurun_ab/libst_$which.so; fi
  python bench.py --steps 24 --warmup 12 --no-cpu-baseline --no-mode-n --headline-only 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']
print('stores $which rep $rep: %.1f xRT, enc %.2f dec %.2f ms/step, pass %.3f ms at %.1f rows, frac %.4f, fc1 %.1f TF/s' % (d['value'], d['phase_ms']['encode_cross_kv'], d['phase_ms']['decode'], r['avg_launch_ms'], r['rows_per_launch'], r['frac'], r['mfma_bound_half']['achieved']))" | tee -a $OUT
done; done
unset SS_LIB_PATH
python bench.py --steps 20 --warmup 4 --no-mode-n > gpurun_out/bench_r06_f_default.json 2> gpurun_out/bench_r06_f_default.err; tail -2 gpurun_out/bench_r06_f_default.err
python - <<'PY' | tee -a $OUT
import json
d = json.loads(open('gpurun_out/bench_r06_f_default.json').read().strip().splitlines()[-1])
r = d['roofline']; print('default line: %.1f xRT, frac %.4f, pass %.3f ms at %.1f rows; mfma half %s; latency %s; host %s; cpu %s' % (d['value'], r['frac'], r['avg_launch_ms'], r['rows_per_launch'], json.dumps(r['mfma_bound_half'])[:600], json.dumps(d['config']['latency'])[:400], json.dumps(d['host_cost'])[:300], json.dumps(d.get('cpu_baseline'))[:500]))
PY
