# round 2, run F: new tests, default bench line, profile of the default configuration, a few more points.   usage: bash tools/gpu_r02_f.sh <tag>
TAG=${1:-r02_f}
mkdir -p gpurun_out
export OMP_WAIT_POLICY=passive
rm -f gpurun_out/parity_report.txt
( time timeout 1800 python -m pytest tests -q -m gpu ) > gpurun_out/pytest_gpu_$TAG.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu_$TAG.log
tail -12 gpurun_out/pytest_gpu_$TAG.log
run() {  # name, env..., args in $ARGS
  name=$1; shift
  env "$@" timeout 600 python bench.py --steps 16 --warmup 8 $ARGS > gpurun_out/bench_${TAG}_$name.json 2> gpurun_out/bench_${TAG}_$name.err
  echo "$name rc=$?"; python - <<PY
import json
try:
    j = json.loads(open("gpurun_out/bench_${TAG}_$name.json").read())
    print({k: j[k] for k in ("value", "ms_per_step", "p50_chunk_latency_ms", "p50_chunk_latency_unloaded_ms", "value_from_host_pcm")}, j["phase_ms"]["encode_cross_kv"], j["phase_ms"]["decode"], {k: j["roofline"][k] for k in ("achieved", "frac", "avg_launch_ms", "passes_overlapping", "rows_per_launch")}, j.get("cpu_baseline", {}).get("value"))
except Exception as e:
    print("ERR", e); print(open("gpurun_out/bench_${TAG}_$name.err").read()[-1500:])
PY
}
ARGS="" run default
ARGS="--no-cpu-baseline --inflight 12 --device-batch 32" run l3_i12_db32 SS_LANES=3
ARGS="--no-cpu-baseline --inflight 12 --device-batch 48" run l2_i12_db48 SS_LANES=2
ARGS="--no-cpu-baseline --inflight 16 --device-batch 64" run l2_i16_db64 SS_LANES=2
ARGS="--no-cpu-baseline --dtype bf16" run default_bf16
PROF_STEPS=8 PROF_WARMUP=8 bash tools/gpu_prof.sh ${TAG} > gpurun_out/prof_${TAG}.txt 2>&1; head -28 gpurun_out/prof_${TAG}.txt | cut -c1-200; tail -3 gpurun_out/prof_${TAG}.txt
