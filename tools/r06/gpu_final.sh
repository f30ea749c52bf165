# Round 6, final numbers: smoke, kernel statistics of the headline region, bench lines f16 (the driver's command) / bf16 / fp8 / base.en bf16
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp OMP_WAIT_POLICY=passive
T=${1:-r06_final}
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
python bench.py > gpurun_out/bench_${T}_f16_default.json 2> gpurun_out/bench_${T}_f16_default.err; echo "default rc=$?"
python bench.py --steps 20 --warmup 4 > gpurun_out/bench_${T}_f16.json 2> gpurun_out/bench_${T}_f16.err; echo "f16 rc=$?"
python bench.py --steps 20 --warmup 4 --dtype bf16 --no-cpu-baseline > gpurun_out/bench_${T}_bf16.json 2> gpurun_out/bench_${T}_bf16.err; echo "bf16 rc=$?"
python bench.py --steps 20 --warmup 4 --dtype fp8 --no-cpu-baseline > gpurun_out/bench_${T}_fp8.json 2> gpurun_out/bench_${T}_fp8.err; echo "fp8 rc=$?"
python bench.py --model base.en --batch 1 --lanes 1 --inflight 1 --device-batch 1 --dtype bf16 --steps 20 --warmup 4 --no-mode-n > gpurun_out/bench_${T}_base.en_b1_bf16.json 2> gpurun_out/bench_${T}_base.en_b1_bf16.err; echo "base.en rc=$?"
python - <<PY
import json
for n in ("f16_default", "f16", "bf16", "fp8", "base.en_b1_bf16"):
    try:
        d = json.loads(open("gpurun_out/bench_${T}_%s.json" % n).read().strip().splitlines()[-1]); r = d["roofline"]; m = r.get("mfma_bound_half", {}); s = d.get("batch8_strict") or {}
        print("%-16s %7.1f xRT  %6.2f ms/step  p50 %6.1f ms  frac %.4f  pass %.3f ms x %.1f rows  fc1 %.1f TF/s (%.3f)  enc alone %s  strict %s x p50 %s  mode_n %s  host %s" % (
            n, d["value"], d["ms_per_step"], d["p50_chunk_latency_ms"], r["frac"], r["avg_launch_ms"], r["rows_per_launch"], m.get("achieved", 0), m.get("frac", 0),
            (m.get("encoder_phase_alone") or {}).get("frac"), s.get("value"), s.get("p50_chunk_latency_ms"), (d.get("mode_n") or {}).get("value"), d["host_cost"]["cores_busy_per_rank_at_this_rate"]))
    except Exception as e:
        print(n, "FAILED", e)
PY
mkdir -p gpurun_out/prof; OUT=$PWD/gpurun_out/prof
( cd /tmp && rocprofv3 --kernel-trace --stats -d $OUT -o ${T} -- python $GRAFT_REPO_ROOT/bench.py --steps 8 --warmup 8 --no-cpu-baseline --no-mode-n --headline-only > $OUT/bench_${T}.log 2> $OUT/bench_${T}.err )
python tools/rocpd_stats.py $(find gpurun_out/prof -name "${T}_results.db" | head -1) gpurun_out/${T}_kernel_stats_f16.md | cut -c1-200 | head -24
rm -f $(find gpurun_out/prof -name "*.db")
