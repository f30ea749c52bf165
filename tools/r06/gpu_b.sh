# Round 6, GPU call B: full GPU suite with every duration; vendor GEMM kernel names / resources; gemm_bench x4 shapes; decoder-weight nt A/B (two builds); configs[1] line
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp OMP_WAIT_POLICY=passive
OUT=gpurun_out/r06_b.txt; : > $OUT
( time timeout 2400 python -m pytest tests -q -m gpu -x --durations=0 ) > gpurun_out/r06_b_pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT
tail -4 gpurun_out/r06_b_pytest.log | tee -a $OUT
echo "== hipBLASLt kernels behind the yardstick (rocprofv3 --kernel-trace --stats)" | tee -a $OUT
mkdir -p gpurun_out/prof_b; P=$PWD/gpurun_out/prof_b
( cd /tmp && rocprofv3 --kernel-trace --stats -d $P -o yard -- python $GRAFT_REPO_ROOT/tools/blaslt_yardstick.py > $P/yard.log 2>&1 )
python - <<'PY' | tee -a $OUT
import sqlite3, glob
for db in glob.glob('gpurun_out/prof_b/**/*.db', recursive=True):
    c = sqlite3.connect(db)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if t.startswith('rocpd_kernel_dispatch')][0]
    ks = [t for t in tabs if t.startswith('rocpd_info_kernel_symbol')][0]
    cols = [r[1] for r in c.execute(f"pragma table_info({kd})")]
    q = f"select s.kernel_name, count(*), avg(d.end-d.start)/1e3, d.workgroup_size_x, d.grid_size_x, d.lds_size, s.arch_vgpr_count, s.accum_vgpr_count, s.sgpr_count, s.group_segment_size from {kd} d join {ks} s on d.kernel_id = s.id group by s.kernel_name order by sum(d.end-d.start) desc limit 14"
    try:
        for r in c.execute(q): print(r)
    except Exception as e:
        print('query failed', e, cols, [r[1] for r in c.execute(f"pragma table_info({ks})")])
PY
rm -f gpurun_out/prof_b/*.db gpurun_out/prof_b/*/*.db
echo "== gemm_bench, the x4 shapes (M = 48000: the 32-window batches the engine runs)" | tee -a $OUT
./tools/gemm_bench.bin 2>&1 | grep -v check | grep -E '^(FC1x4|FC2x4|Ox4|QKx4) ' | tee -a $OUT
echo "== decoder weight fragments as nt loads (variant build) vs plain, alternating" | tee -a $OUT
for rep in 1 2; do for which in plain nt; do
  if [ $which = nt ]; then export SS_LIB_PATH=$PWD/gpurun_ab/libweights_nt.so; else unset SS_LIB_PATH; fi
  python bench.py --steps 24 --warmup 12 --no-cpu-baseline --no-mode-n --headline-only 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']
print('weights $which rep $rep: %.1f xRT, pass %.3f ms at %.1f rows, frac %.4f' % (d['value'], r['avg_launch_ms'], r['rows_per_launch'], r['frac']))" | tee -a $OUT
done; done
unset SS_LIB_PATH
echo "== configs[1]: base.en, one chunk at a time, bf16 and f16" | tee -a $OUT
for dt in bf16 f16; do
python bench.py --model base.en --batch 1 --lanes 1 --inflight 1 --device-batch 1 --dtype $dt --steps 20 --warmup 4 --no-mode-n > gpurun_out/bench_r06_b_base.en_b1_$dt.json 2> gpurun_out/bench_r06_b_base.en_b1_$dt.err
python -c "
import json
d = json.loads(open('gpurun_out/bench_r06_b_base.en_b1_$dt.json').read().strip().splitlines()[-1]); r = d['roofline']
print('base.en B=1 $dt: %.1f xRT, %.2f ms per chunk, p50 %.1f ms, pass %.4f ms, frac %.4f' % (d['value'], d['ms_per_step'], d['p50_chunk_latency_ms'], r['avg_launch_ms'], r['frac']))" | tee -a $OUT
done
