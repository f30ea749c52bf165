# A/B of the 256 x 256 GEMM main loop on the headline: 32-deep stages (SS_GEMM_K64=0) vs 64-deep stages (default), same box, alternating
cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do for k in 0 1; do
  SS_GEMM_K64=$k python bench.py --steps 24 --warmup 8 --no-cpu-baseline --no-mode-n --headline-only 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('TK %d rep $rep: %.1f xRT, %.2f ms per step, encoder phase %.2f ms, decode phase %.2f ms' % (64 if $k else 32, d['value'], d['ms_per_step'], d['phase_ms']['encode_cross_kv'], d['phase_ms']['decode']))"
done; done
