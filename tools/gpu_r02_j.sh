# round 2, run J: unsplit cross-attention for wide passes (A/B), bf16 oracle fix.   usage: bash tools/gpu_r02_j.sh <tag>
TAG=${1:-r02_j}
mkdir -p gpurun_out
export OMP_WAIT_POLICY=passive
rm -f gpurun_out/parity_report.txt
( time timeout 1800 python -m pytest tests -q -m gpu ) > gpurun_out/pytest_gpu_$TAG.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu_$TAG.log
tail -8 gpurun_out/pytest_gpu_$TAG.log; grep bf16 gpurun_out/parity_report.txt | cut -c1-250
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
ARGS="" run split SS_CROSS_DIRECT_PAIRS=1000000
ARGS="" run direct320
ARGS="" run direct160 SS_CROSS_DIRECT_PAIRS=160
ARGS="--inflight 16 --device-batch 64" run direct320_db64
ARGS="--inflight 4 --device-batch 16" run direct320_db16
ARGS="--inflight 1 --device-batch 8" run direct160_b8 SS_CROSS_DIRECT_PAIRS=160 SS_LANES=1
