# Up to how many rows do the LayerNorm-prologue GEMVs pay?  (dev switch SS_LN_FUSE_ROWS; the product dispatches them for <= kLnFuseRows = 4 rows)
# large-v3, one batch of 8 chunks at a time on one lane (8 rows per pass: `batch8_strict`'s configuration) and 2 / 4 chunks at a time.
cd "$GRAFT_REPO_ROOT" || exit 1
for b in 8 4 2; do for r in 4 8 16; do
  [ $r -lt $b ] && [ $r -ne 4 ] && continue
  line=$(SS_LN_FUSE_ROWS=$r python bench.py --no-cpu-baseline --headline-only --batch $b --lanes 1 --device-batch $b --inflight 1 --steps 12 --warmup 3 2>/tmp/ab.err) || { echo "batch $b rows<=$r FAILED"; tail -2 /tmp/ab.err; continue; }
  python -c "import json,sys; j=json.loads(sys.argv[1]); print('batch %d, LN prologues for <= %2d rows: %7.1f xRT, %6.1f ms per batch, pass %.3f ms x %.1f rows' % ($b, $r, j['value'], j['ms_per_step'], j['roofline']['avg_launch_ms'], j['roofline']['rows_per_launch']))" "$line"
done; done
