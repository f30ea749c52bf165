# Round 6, GPU call N: the de-phased GEMM tile schedule: self-tests (bitwise vs the plain schedule), gemm_bench A/B/A at sustained load, engine A/B
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp OMP_WAIT_POLICY=passive
OUT=gpurun_out/r06_n_gemm_dephase.txt; : > $OUT
( timeout 900 python -m pytest tests/test_gpu_gemm.py -q -m gpu -x 2>&1 | tail -5 ) | tee -a $OUT
for D in 0 3 0 3; do echo "-- SS_GEMM_DEPHASE=$D (40 launches per line)" | tee -a $OUT
  SS_GEMM_DEPHASE=$D SS_GEMM_REPS=40 ./tools/gemm_bench.bin 2>&1 | grep -E ' (store|gelu|res_f32|v\^T) ' | grep -E '^(FC1x4|FC2x4|Ox4|QKx4|FC1|FC2|QK|O|crossKV) ' | tee -a $OUT; done
echo "== engine, alternating SS_GEMM_DEPHASE=0 / 3" | tee -a $OUT
for rep in 1 2; do for D in 0 3; do
  SS_GEMM_DEPHASE=$D python bench.py --steps 24 --warmup 12 --no-cpu-baseline --no-mode-n --headline-only 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']
print('dephase $D rep $rep: %.1f xRT, enc %.2f dec %.2f ms/step, pass %.3f ms at %.1f rows, frac %.4f, fc1 %.1f TF/s' % (d['value'], d['phase_ms']['encode_cross_kv'], d['phase_ms']['decode'], r['avg_launch_ms'], r['rows_per_launch'], r['frac'], r['mfma_bound_half']['achieved']))" | tee -a $OUT
done; done
for D in 0 3; do
  SS_GEMM_DEPHASE=$D python bench.py --batch 8 --lanes 1 --inflight 1 --device-batch 8 --steps 6 --warmup 2 --no-cpu-baseline --no-mode-n --headline-only 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('dephase $D, one batch of 8 at a time: %.1f xRT, enc %.2f ms per 8 windows' % (d['value'], d['phase_ms']['encode_cross_kv']))" | tee -a $OUT
done
