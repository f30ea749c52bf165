# Round 6, GPU call Q: windows per encoder pass (SS_ENC_MAX_WINDOWS): do 8 - 16-window GEMM launches (activations MALL-resident, shorter kernels beside the other lanes' chains) beat 32-window ones in the pipeline?
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp OMP_WAIT_POLICY=passive
OUT=gpurun_out/r06_q_enc_windows_per_pass.txt; : > $OUT
for rep in 1 2; do for W in 0 8 12 16; do
  if [ $W = 0 ]; then unset SS_ENC_MAX_WINDOWS; else export SS_ENC_MAX_WINDOWS=$W; fi
  python bench.py --steps 24 --warmup 12 --no-cpu-baseline --no-mode-n --headline-only 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']
print('windows per encoder pass <= $W rep $rep: %.1f xRT, p50 %.1f ms, enc %.2f dec %.2f ms/step, pass %.3f ms at %.1f rows, frac %.4f' % (d['value'], d['p50_chunk_latency_ms'], d['phase_ms']['encode_cross_kv'], d['phase_ms']['decode'], r['avg_launch_ms'], r['rows_per_launch'], r['frac']))" | tee -a $OUT
done; done
