# Round 6, GPU call A: baseline on this box, nt loads on the cross-K/V stream (A/B/A/B), rows-per-pass sweep with the z-split GEMVs, vendor GEMM yardstick A/B/A.
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp OMP_WAIT_POLICY=passive
OUT=gpurun_out/r06_a_ab.txt; : > $OUT
sum() { python -c "
import sys, json
lab = sys.argv[1]
try:
    d = json.loads(sys.stdin.read().strip().splitlines()[-1])
    r = d['roofline']; s = d.get('steady_state') or {}
    print('%-52s %7.1f xRT  steady %7.1f  p50 %7.1f ms  enc %6.2f dec %6.2f ms/step  pass %.3f ms x %.1f rows  frac %.4f' % (lab, d['value'], s.get('value') or 0, d['p50_chunk_latency_ms'], d['phase_ms']['encode_cross_kv'], d['phase_ms']['decode'], r['avg_launch_ms'], r['rows_per_launch'], r['frac']))
except Exception as e:
    print('%-52s FAILED %s' % (lab, e))
" "$1" | tee -a $OUT; }
run() {  # <label> <lanes> <device-batch> <inflight> <steps> [extra args]
  lab=$1; shift; lanes=$1; shift; db=$1; shift; inf=$1; shift; steps=$1; shift
  python bench.py --no-cpu-baseline --no-mode-n --headline-only --steps $steps --warmup $inf --lanes $lanes --device-batch $db --inflight $inf "$@" 2>/tmp/ab.err | sum "$lab"
}
( timeout 900 python -m pytest tests/test_gpu_batch_invariance.py "tests/test_gpu_bench_config.py::test_decoder_pass_32_64_and_128_rows_vs_oracle" -q -m gpu -x 2>&1 | tail -5 ) | tee -a $OUT
echo "== full default line (with batch8_strict)" | tee -a $OUT
python bench.py --steps 20 --warmup 4 --no-cpu-baseline --no-mode-n > gpurun_out/bench_r06_a_default.json 2> gpurun_out/bench_r06_a_default.err
python - <<'PY' | tee -a $OUT
import json
d = json.loads(open('gpurun_out/bench_r06_a_default.json').read().strip().splitlines()[-1])
s = d.get('batch8_strict') or {}
print('default: %.1f xRT p50 %.1f ms frac %.4f pass %.3f ms; strict %.1f xRT p50 %.1f ms frac %.4f pass %.3f ms; unloaded %.1f ms; fc1 %.1f TF/s' % (
  d['value'], d['p50_chunk_latency_ms'], d['roofline']['frac'], d['roofline']['avg_launch_ms'], s.get('value', 0), s.get('p50_chunk_latency_ms', 0),
  s.get('roofline', {}).get('frac', 0), s.get('roofline', {}).get('avg_launch_ms', 0), d.get('p50_chunk_latency_unloaded_ms') or 0, d['phase_roofline']['encoder_fc1_gemm']['achieved']))
PY
echo "== SS_CROSS_NT A/B/A/B (3 lanes x 32 rows x 12 in flight, --steps 24)" | tee -a $OUT
for rep in 1 2; do for nt in 0 1; do SS_CROSS_NT=$nt run "cross nt=$nt rep $rep" 3 32 12 24; done; done
echo "== SS_CROSS_NT on batch8_strict-like (1 lane x 8 rows, 1 in flight)" | tee -a $OUT
for nt in 0 1; do SS_CROSS_NT=$nt run "strict nt=$nt" 1 8 1 6; done
echo "== rows per pass with z-split GEMVs (SS_GEMV_ZSPLIT=1) vs the CT=4/8 form (=0)" | tee -a $OUT
for z in 1 0; do
  export SS_GEMV_ZSPLIT=$z
  run "z=$z 2 lanes x  64 rows, 128 in flight" 2 64 16 48
  run "z=$z 1 lane  x 128 rows, 128 in flight" 1 128 16 48
  run "z=$z 3 lanes x  64 rows, 192 in flight" 3 64 24 72
done
export SS_GEMV_ZSPLIT=1
run "z=1 2 lanes x  48 rows,  96 in flight" 2 48 12 36
run "z=1 1 lane  x  96 rows,  96 in flight" 1 96 12 36
run "z=1 2 lanes x  96 rows, 192 in flight" 2 96 24 72
run "z=1 2 lanes x 128 rows, 256 in flight" 2 128 32 96
run "z=1 3 lanes x  32 rows,  96 in flight" 3 32 12 36
unset SS_GEMV_ZSPLIT
echo "== vendor yardstick A/B/A (same operand distributions: A uniform [-1,1), W uniform [-0.05,0.05))" | tee -a $OUT
./tools/gemm_bench.bin 2>&1 | grep -v check | grep -E 'store|res_f32|gelu|v\^T' | grep -E '^(FC1|FC2|QKV|QK|O|crossKV) ' | tee -a $OUT
python tools/blaslt_yardstick.py 2>&1 | tee -a $OUT
./tools/gemm_bench.bin 2>&1 | grep -v check | grep -E 'store|res_f32|gelu|v\^T' | grep -E '^(FC1|FC2|QKV|QK|O|crossKV) ' | tee -a $OUT
