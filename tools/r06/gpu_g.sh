# Round 6, GPU call G: the two-level split-K ticket + tile-statistics LayerNorm experiment, libln_ticket.so vs the product build
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp OMP_WAIT_POLICY=passive
OUT=gpurun_out/r06_g_ln_ticket.txt; : > $OUT
echo "== correctness of the experiment build against the oracle (tolerance tests of the wide passes)" | tee -a $OUT
( SS_LIB_PATH=$PWD/gpurun_ab/libln_ticket.so timeout 1200 python -m pytest "tests/test_gpu_bench_config.py::test_decoder_pass_32_64_and_128_rows_vs_oracle" "tests/test_gpu_bench_config.py::test_bench_engine_32_row_passes_vs_oracle" tests/test_gpu_parity.py -q -m gpu -k "not ladder and not openai" 2>&1 | tail -6 ) | tee -a $OUT
sum() { python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']
print('%-34s %7.1f xRT  p50 %6.1f ms  enc %6.2f dec %6.2f ms/step  pass %.3f ms x %.1f rows  frac %.4f' % (sys.argv[1], d['value'], d['p50_chunk_latency_ms'], d['phase_ms']['encode_cross_kv'], d['phase_ms']['decode'], r['avg_launch_ms'], r['rows_per_launch'], r['frac']))" "$1" | tee -a $OUT; }
echo "== headline configuration (3 lanes x 32 rows, 96 in flight), alternating builds" | tee -a $OUT
for rep in 1 2; do for which in product ticket; do
  if [ $which = ticket ]; then export SS_LIB_PATH=$PWD/gpurun_ab/libln_ticket.so; else unset SS_LIB_PATH; fi
  python bench.py --steps 24 --warmup 12 --no-cpu-baseline --no-mode-n --headline-only 2>/dev/null | sum "$which rep $rep"
done; done
echo "== one lane, one pass at a time: 32 rows and 8 rows (batch8_strict's shape)" | tee -a $OUT
for which in product ticket product ticket; do
  if [ $which = ticket ]; then export SS_LIB_PATH=$PWD/gpurun_ab/libln_ticket.so; else unset SS_LIB_PATH; fi
  python bench.py --batch 32 --lanes 1 --inflight 1 --device-batch 32 --steps 6 --warmup 2 --no-cpu-baseline --no-mode-n --headline-only 2>/dev/null | sum "$which 1 lane x 32 rows"
  python bench.py --batch 8 --lanes 1 --inflight 1 --device-batch 8 --steps 6 --warmup 2 --no-cpu-baseline --no-mode-n --headline-only 2>/dev/null | sum "$which 1 lane x 8 rows"
done
echo "== per-kernel durations, one lane x 32 rows (rocprofv3 --kernel-trace --stats)" | tee -a $OUT
mkdir -p gpurun_out/prof_g; P=$PWD/gpurun_out/prof_g
for which in product ticket; do
  if [ $which = ticket ]; then export SS_LIB_PATH=$PWD/gpurun_ab/libln_ticket.so; else unset SS_LIB_PATH; fi
  ( cd /tmp && rocprofv3 --kernel-trace --stats -d $P -o $which -- python $GRAFT_REPO_ROOT/bench.py --batch 32 --lanes 1 --inflight 1 --device-batch 32 --steps 3 --warmup 1 --no-cpu-baseline --no-mode-n --headline-only > $P/$which.log 2>&1 )
  DB=$(find $P -name "${which}_results.db" | head -1)
  echo "-- $which" | tee -a $OUT
  python tools/rocpd_stats.py $DB gpurun_out/r06_g_kernel_stats_one_lane_$which.md | cut -c1-190 | head -16 | tee -a $OUT
done
rm -rf gpurun_out/prof_g
