# Round 6, GPU call O: cache policy of the GEMM's operand DMA (A = activations, W = weights; 2 = non-temporal): gemm_bench and the engine, alternating
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp OMP_WAIT_POLICY=passive
OUT=gpurun_out/r06_o_gemm_operand_policy.txt; : > $OUT
for B in gemm_bench gemm_bench_a2w0 gemm_bench_a0w2 gemm_bench_a2w2 gemm_bench gemm_bench_a2w0; do echo "-- $B (40 launches per line)" | tee -a $OUT
  SS_GEMM_REPS=40 ./tools/$B.bin 2>&1 | grep -E ' (store|gelu|res_f32) ' | grep -E '^(FC1x4|FC2x4|Ox4|QKx4|FC1|QK|crossKV) ' | tee -a $OUT; done
echo "== engine, alternating builds" | tee -a $OUT
for rep in 1 2; do for which in plain a2w0 a2w2; do
  if [ $which = plain ]; then unset SS_LIB_PATH; else export SS_LIB_PATH=$PWD/gpurun_ab/libgemm_$which.so; fi
  python bench.py --steps 24 --warmup 12 --no-cpu-baseline --no-mode-n --headline-only 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']
print('operands $which rep $rep: %.1f xRT, enc %.2f dec %.2f ms/step, pass %.3f ms at %.1f rows, frac %.4f, fc1 %.1f TF/s' % (d['value'], d['phase_ms']['encode_cross_kv'], d['phase_ms']['decode'], r['avg_launch_ms'], r['rows_per_launch'], r['frac'], r['mfma_bound_half']['achieved']))" | tee -a $OUT
done; done
unset SS_LIB_PATH
( timeout 300 python -m pytest tests/test_gpu_variants.py -q -m gpu -k "whisper_h_full_surface" 2>&1 | tail -3 ) | tee -a $OUT
