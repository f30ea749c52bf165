# round 2, run I: full GPU suite, default bench (full-depth cpu_baseline), BASELINE configs[1] (base.en bf16), configs[3]-shaped stream bench.   usage: bash tools/gpu_r02_i.sh <tag>
TAG=${1:-r02_i}
mkdir -p gpurun_out
export OMP_WAIT_POLICY=passive
rm -f gpurun_out/parity_report.txt
( time timeout 1800 python -m pytest tests -q -m gpu ) > gpurun_out/pytest_gpu_$TAG.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu_$TAG.log
tail -8 gpurun_out/pytest_gpu_$TAG.log
( time timeout 900 python bench.py ) > gpurun_out/bench_${TAG}_default.json 2> gpurun_out/bench_${TAG}_default.err; echo "bench rc=$?"; python - <<PY
import json
j = json.loads(open("gpurun_out/bench_${TAG}_default.json").read().splitlines()[0])
print({k: j[k] for k in ("value", "ms_per_step", "p50_chunk_latency_ms", "p50_chunk_latency_unloaded_ms")}, j["roofline"]["frac"], j["roofline"]["traffic"], j["cpu_baseline"])
PY
tail -4 gpurun_out/bench_${TAG}_default.err
for dt in bf16 f16; do
  timeout 600 python bench.py --model base.en --batch 1 --inflight 1 --device-batch 1 --dtype $dt --no-cpu-baseline --steps 20 --warmup 5 > gpurun_out/bench_${TAG}_base.en_b1_$dt.json 2>/dev/null
  cut -c1-330 gpurun_out/bench_${TAG}_base.en_b1_$dt.json; echo
  timeout 600 python bench.py --model base.en --batch 8 --dtype $dt --no-cpu-baseline > gpurun_out/bench_${TAG}_base.en_b8_$dt.json 2>/dev/null
  cut -c1-330 gpurun_out/bench_${TAG}_base.en_b8_$dt.json; echo
done
for mb in 8 16 32; do
  timeout 600 python tools/stream_bench.py --streams 64 --seconds 30 --max-batch $mb 2>/dev/null | tee -a gpurun_out/stream_bench_$TAG.jsonl
done
timeout 600 python tools/stream_bench.py --streams 8 --seconds 30 --max-batch 8 2>/dev/null | tee -a gpurun_out/stream_bench_$TAG.jsonl
