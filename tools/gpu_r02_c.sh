# round 2, run C: decode step v2 (narrow tiles) vs v1, lanes 1/2/3.   usage: bash tools/gpu_r02_c.sh <tag>
TAG=${1:-r02_c}
mkdir -p gpurun_out
export OMP_WAIT_POLICY=passive
rm -f gpurun_out/parity_report.txt
( time timeout 1800 python -m pytest tests -q -m gpu ) > gpurun_out/pytest_gpu_$TAG.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu_$TAG.log
tail -12 gpurun_out/pytest_gpu_$TAG.log
cat gpurun_out/parity_report.txt | cut -c1-250
run() {  # name, env..., args
  name=$1; shift
  env "$@" timeout 600 python bench.py --steps 6 --warmup 2 --no-cpu-baseline $ARGS > gpurun_out/bench_${TAG}_$name.json 2> gpurun_out/bench_${TAG}_$name.err
  echo "$name rc=$?"; python - <<PY
import json
try:
    j = json.loads(open("gpurun_out/bench_${TAG}_$name.json").read())
    print({k: j[k] for k in ("value", "ms_per_step", "p50_chunk_latency_ms")}, j["phase_ms"]["encode_cross_kv"], j["phase_ms"]["decode"], {k: j["roofline"][k] for k in ("achieved", "frac", "avg_launch_ms", "passes_overlapping")})
except Exception as e:
    print("ERR", e); print(open("gpurun_out/bench_${TAG}_$name.err").read()[-1500:])
PY
}
ARGS="--inflight 1" run v1_l1 SS_LANES=1 SS_DECODE_V1=1
ARGS="--inflight 1" run v2_l1 SS_LANES=1
ARGS="--inflight 2" run v1_l2 SS_LANES=2 SS_DECODE_V1=1
ARGS="--inflight 2" run v2_l2 SS_LANES=2
ARGS="--inflight 3" run v2_l3 SS_LANES=3
ARGS="--inflight 2 --batch 16" run v2_l2_b16 SS_LANES=2
bash tools/gpu_prof.sh ${TAG} > gpurun_out/prof_${TAG}.txt 2>&1; head -24 gpurun_out/prof_${TAG}.txt | cut -c1-200; tail -3 gpurun_out/prof_${TAG}.txt
