# round 2, run E: 17..64-row fused decoder passes; device-batch / lanes sweep.   usage: bash tools/gpu_r02_e.sh <tag>
TAG=${1:-r02_e}
mkdir -p gpurun_out
export OMP_WAIT_POLICY=passive
rm -f gpurun_out/parity_report.txt
( time timeout 1800 python -m pytest tests -q -m gpu ) > gpurun_out/pytest_gpu_$TAG.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu_$TAG.log
tail -12 gpurun_out/pytest_gpu_$TAG.log
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
ARGS="--inflight 4 --device-batch 32" run l1_i4_db32 SS_LANES=1
ARGS="--inflight 4 --device-batch 32" run l1_i4_db32_old SS_LANES=1 SS_DECODE_WIDE=0
ARGS="--inflight 8 --device-batch 32" run l2_i8_db32 SS_LANES=2
ARGS="--inflight 8 --device-batch 64" run l1_i8_db64 SS_LANES=1
ARGS="--inflight 16 --device-batch 64" run l2_i16_db64 SS_LANES=2
ARGS="--inflight 4 --device-batch 16" run l2_i4_db16 SS_LANES=2
ARGS="--inflight 6 --device-batch 24" run l2_i6_db24 SS_LANES=2
