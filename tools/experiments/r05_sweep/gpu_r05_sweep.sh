# gpurun -- 'bash tools/gpu_r05_sweep.sh': lanes x rows per pass x steps in flight around the bench default, headline region only (profiles/r05_r_lanes_rows_sweep.txt)
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
OUT=gpurun_out/r05_r_lanes_rows_sweep.txt
echo "lanes device_batch inflight | xRT (--steps 20 --warmup 4)  steady  p50_ms  pass_ms rows/pass frac" > $OUT
for cfg in "3 32 12" "3 40 15" "3 36 14" "4 24 12" "2 48 12" "3 32 16" "3 28 11" "3 32 12"; do
  set -- $cfg
  python bench.py --steps 20 --warmup 4 --lanes $1 --device-batch $2 --inflight $3 --no-cpu-baseline --no-mode-n > /tmp/b.json 2>/dev/null
  python - "$cfg" >> $OUT <<'PY'
import json, sys
d = json.load(open('/tmp/b.json'))
print(sys.argv[1], '|', d['value'], (d.get('steady_state') or {}).get('value'), d['p50_chunk_latency_ms'], d['roofline']['avg_launch_ms'], d['roofline']['rows_per_launch'], d['roofline']['frac'])
PY
done
cat $OUT
