cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
run() { name=$1; shift; env "$@" timeout 300 python bench.py --no-cpu-baseline $ARGS 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.read()); print('$name', j['value'], j['p50_chunk_latency_ms'], j['roofline']['avg_launch_ms'], j['roofline']['rows_per_launch'], j['roofline']['frac'], j['roofline']['passes_overlapping'])"; }
ARGS="--inflight 16" run f16_l4_db32_i16 SS_LANES=4
ARGS="--inflight 9 --device-batch 24" run f16_l3_db24_i9 SS_LANES=3
ARGS="--inflight 8 --device-batch 16" run f16_l4_db16_i8 SS_LANES=4
ARGS="--inflight 10" run f16_l3_db32_i10 SS_LANES=3
ARGS="--inflight 12" run f16_l3_db32_i12 SS_LANES=3
ARGS="--inflight 8" run f16_l3_db32_i8 SS_LANES=3
ARGS="--inflight 20 --device-batch 40" run f16_l4_db40_i20 SS_LANES=4
ARGS="--dtype fp8 --inflight 16" run fp8_l4_db32_i16 SS_LANES=4
