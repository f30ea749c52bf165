# round 2, run B: GPU suite after the lanes refactor + lanes / in-flight sweep of the bench.   usage: bash tools/gpu_r02_b.sh <tag>
TAG=${1:-r02_b}
mkdir -p gpurun_out
export OMP_WAIT_POLICY=passive
rm -f gpurun_out/parity_report.txt
( time timeout 1800 python -m pytest tests -q -m gpu -x ) > gpurun_out/pytest_gpu_$TAG.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu_$TAG.log
tail -15 gpurun_out/pytest_gpu_$TAG.log
cat gpurun_out/parity_report.txt
for cfg in "1 1" "2 2" "3 3" "4 4" "2 3"; do
  set -- $cfg
  SS_LANES=$1 timeout 600 python bench.py --steps 6 --warmup 2 --inflight $2 --no-cpu-baseline > gpurun_out/bench_${TAG}_l$1_i$2.json 2> gpurun_out/bench_${TAG}_l$1_i$2.err
  echo "lanes $1 inflight $2 rc=$?"; python - <<PY
import json
try:
    j = json.loads(open("gpurun_out/bench_${TAG}_l$1_i$2.json").read())
    print({k: j[k] for k in ("value", "ms_per_step", "p50_chunk_latency_ms")}, j["phase_ms"], {k: j["roofline"][k] for k in ("achieved", "frac", "avg_launch_ms", "passes_overlapping")})
except Exception as e:
    print("ERR", e); print(open("gpurun_out/bench_${TAG}_l$1_i$2.err").read()[-1500:])
PY
done
