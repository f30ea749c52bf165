# round 2, run H: driver-style step counts (--steps 20 --warmup 5) over in-flight / device-batch choices; continuous-batching test; PMC.   usage: bash tools/gpu_r02_h.sh <tag>
TAG=${1:-r02_h}
mkdir -p gpurun_out
export OMP_WAIT_POLICY=passive
timeout 900 python -m pytest tests -q -m gpu -k "continuous or pool or async or long_prompt" 2>&1 | tail -5
run() {  # name, args in $ARGS
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
ARGS="--steps 20 --warmup 5" run k20_i8_db32
ARGS="--steps 20 --warmup 5 --inflight 4 --device-batch 16" run k20_i4_db16
ARGS="--steps 20 --warmup 5 --inflight 8 --device-batch 16" run k20_i8_db16
ARGS="--steps 20 --warmup 5 --inflight 6 --device-batch 24" run k20_i6_db24
ARGS="--steps 20 --warmup 5 --inflight 5 --device-batch 20" run k20_i5_db20
ARGS="--steps 5 --warmup 2" run k5_i8_db32
ARGS="--steps 5 --warmup 2 --inflight 4 --device-batch 16" run k5_i4_db16
ARGS="" run default_k24
bash tools/gpu_pmc.sh $TAG > gpurun_out/pmc_$TAG.txt 2>&1; tail -6 gpurun_out/pmc_$TAG.txt | cut -c1-220
