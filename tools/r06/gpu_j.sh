# Round 6, GPU call J: the driver's command (--steps 20 --warmup 5) at operating points whose rounds divide the 160 timed chunks evenly; GEMV plans at 32 rows
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp OMP_WAIT_POLICY=passive
OUT=gpurun_out/r06_j.txt; : > $OUT
sum() { python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']; s = d.get('steady_state') or {}
print('%-44s %7.1f xRT  steady %7.1f  p50 %6.1f ms  pass %.3f ms x %.1f rows  frac %.4f' % (sys.argv[1], d['value'], s.get('value') or 0, d['p50_chunk_latency_ms'], r['avg_launch_ms'], r['rows_per_launch'], r['frac']))" "$1" | tee -a $OUT; }
for rep in 1 2; do
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-mode-n --inflight 12 --device-batch 32 2>/dev/null | sum "K=20 W=5: 12 in flight, batches <= 32"
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-mode-n --inflight 10 --device-batch 27 2>/dev/null | sum "K=20 W=5: 10 in flight, batches <= 27"
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-mode-n --inflight 10 --device-batch 32 2>/dev/null | sum "K=20 W=5: 10 in flight, batches <= 32"
done
python bench.py --steps 24 --warmup 8 --no-cpu-baseline --no-mode-n --inflight 12 --device-batch 32 2>/dev/null | sum "K=24 W=8: 12 in flight, batches <= 32"
echo "== decode GEMV plans at 32 rows (independent launches, cold weights)" | tee -a $OUT
SS_GEMV_M=32 ./tools/gemv_bench.bin 2>&1 | grep -v logits | tee -a $OUT
