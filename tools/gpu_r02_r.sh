# round 2, run R: lanes / batch sweep for the fp8 engine (its cross stream is half the f16 engine's) and two f16 points
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
run() { name=$1; shift; env "$@" timeout 300 python bench.py --no-cpu-baseline $ARGS 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.read()); print('$name', j['value'], j['p50_chunk_latency_ms'], j['roofline']['avg_launch_ms'], j['roofline']['rows_per_launch'], j['roofline']['frac'])"; }
ARGS="--dtype fp8" run fp8_l2_db32 SS_LANES=2
ARGS="--dtype fp8" run fp8_l3_db32 SS_LANES=3
ARGS="--dtype fp8 --inflight 12" run fp8_l3_db32_i12 SS_LANES=3
ARGS="--dtype fp8 --inflight 12 --device-batch 48" run fp8_l2_db48_i12 SS_LANES=2
ARGS="--dtype fp8 --inflight 16 --device-batch 64" run fp8_l2_db64_i16 SS_LANES=2
ARGS="--dtype fp8 --inflight 8 --device-batch 64" run fp8_l1_db64_i8 SS_LANES=1
ARGS="--dtype fp8 --inflight 4 --device-batch 16" run fp8_l2_db16_i4 SS_LANES=2
ARGS="--inflight 12" run f16_l3_db32_i12 SS_LANES=3
ARGS="--inflight 16 --device-batch 64" run f16_l2_db64_i16 SS_LANES=2
