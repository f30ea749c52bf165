# gpurun -- 'bash tools/experiments/r05_sweep/gpu_r05_cbstart.sh': SS_CB_START_MIN (windows that must be waiting before a running group pauses its decoders for their
# encoder pass) against the natural-EOT leg of the bench (mode_n: reference parameters, 96 / 64 / 32 chunks in flight)
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
OUT=gpurun_out/r05_s_cb_start_min_sweep.txt
echo "SS_CB_START_MIN | mode_n operating points (chunks in flight, audio-s/s, p50 ms) ; rows per pass ; windows started midway" > $OUT
for v in 4 2 8 12 16 4; do
  SS_CB_START_MIN=$v python bench.py --steps 4 --warmup 2 --no-cpu-baseline > /tmp/b.json 2>/dev/null
  python - "$v" >> $OUT <<'PY'
import json, sys
d = json.load(open('/tmp/b.json'))['mode_n']
print(sys.argv[1], '|', [(p['chunks_in_flight'], p['value'], p['p50_chunk_latency_ms']) for p in d['operating_points']], ';', d['rows_per_pass'], ';', d['windows_started_midway'])
PY
done
cat $OUT
