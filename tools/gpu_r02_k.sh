# round 2, run K: token-level continuous batching (event-driven window loop): full suite, smoke, default bench, latency modes.   usage: bash tools/gpu_r02_k.sh <tag>
TAG=${1:-r02_k}
mkdir -p gpurun_out
export OMP_WAIT_POLICY=passive
rm -f gpurun_out/parity_report.txt
( time timeout 1800 python -m pytest tests -q -m gpu ) > gpurun_out/pytest_gpu_$TAG.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu_$TAG.log
tail -30 gpurun_out/pytest_gpu_$TAG.log | cut -c1-250
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
run() {  # name, env..., args in $ARGS
  name=$1; shift
  env "$@" timeout 600 python bench.py --no-cpu-baseline $ARGS > gpurun_out/bench_${TAG}_$name.json 2> gpurun_out/bench_${TAG}_$name.err
  echo "$name rc=$?"; python - <<PY
import json
try:
    j = json.loads(open("gpurun_out/bench_${TAG}_$name.json").read())
    print({k: j[k] for k in ("value", "ms_per_step", "p50_chunk_latency_ms", "p50_chunk_latency_unloaded_ms")}, {k: j["roofline"][k] for k in ("achieved", "frac", "avg_launch_ms", "passes_overlapping", "rows_per_launch")})
except Exception as e:
    print("ERR", e); print(open("gpurun_out/bench_${TAG}_$name.err").read()[-1500:])
PY
}
ARGS="" run default
ARGS="--inflight 1 --device-batch 8" run b8_alone SS_LANES=1
ARGS="--fixed-steps 0 --steps 6 --warmup 2" run modeN
ARGS="--fixed-steps 0 --steps 6 --warmup 2" run modeN_start1 SS_CB_START_MIN=1
ARGS="--fixed-steps 0 --steps 6 --warmup 2" run modeN_start8 SS_CB_START_MIN=64
