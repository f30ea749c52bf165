# round 2, run D: how many rows per decoder pass / passes in flight pay off (v1 step).   usage: bash tools/gpu_r02_d.sh <tag>
TAG=${1:-r02_d}
mkdir -p gpurun_out
export OMP_WAIT_POLICY=passive
run() {  # name, env..., args in $ARGS
  name=$1; shift
  env "$@" timeout 600 python bench.py --steps 8 --warmup 4 --no-cpu-baseline $ARGS > gpurun_out/bench_${TAG}_$name.json 2> gpurun_out/bench_${TAG}_$name.err
  echo "$name rc=$?"; python - <<PY
import json
try:
    j = json.loads(open("gpurun_out/bench_${TAG}_$name.json").read())
    print({k: j[k] for k in ("value", "ms_per_step", "p50_chunk_latency_ms")}, j["phase_ms"]["encode_cross_kv"], j["phase_ms"]["decode"], {k: j["roofline"][k] for k in ("achieved", "frac", "avg_launch_ms", "passes_overlapping", "rows_per_launch")})
except Exception as e:
    print("ERR", e); print(open("gpurun_out/bench_${TAG}_$name.err").read()[-1500:])
PY
}
ARGS="--inflight 2" run l2_i2_db8 SS_LANES=2
ARGS="--inflight 2 --device-batch 16" run l1_i2_db16 SS_LANES=1
ARGS="--inflight 4 --device-batch 16" run l2_i4_db16 SS_LANES=2
ARGS="--inflight 4 --device-batch 32" run l1_i4_db32 SS_LANES=1
ARGS="--inflight 8 --device-batch 32" run l2_i8_db32 SS_LANES=2
ARGS="--inflight 6 --device-batch 16" run l3_i6_db16 SS_LANES=3
timeout 900 python -m pytest tests -q -m gpu -k "pool or batch8 or async" 2>&1 | tail -3
