# Round 6, GPU call C: new / changed tests; vendor GEMM kernel names + resources; residual-epilogue prefetch depth A/B (gemm_bench builds a1..a4)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp OMP_WAIT_POLICY=passive
OUT=gpurun_out/r06_c.txt; : > $OUT
echo "nproc $(nproc), mem $(free -g | awk '/Mem/{print $2}') GB" | tee -a $OUT
( time timeout 1500 python -m pytest tests/test_gpu_variants.py tests/test_gpu_batch_invariance.py tests/test_gpu_audio_ctx.py tests/test_gpu_multi.py "tests/test_gpu_lifetime.py::test_engine_create_refuses_a_configuration_that_cannot_fit" tests/test_gpu_parity.py -q -m gpu --durations=12 ) > gpurun_out/r06_c_pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT
tail -22 gpurun_out/r06_c_pytest.log | cut -c1-250 | tee -a $OUT
echo "== residual epilogue: operand row groups in flight (SS_RES_AHEAD 1 = rounds 2-5), A/B/A" | tee -a $OUT
for A in 1 2 3 4 1 3; do echo "-- ahead $A" | tee -a $OUT; ./tools/gemm_bench_a$A.bin 2>&1 | grep -E 'res_f32' | grep -E '^(FC2|O|FC2x4|Ox4|QKx4) ' | tee -a $OUT; done
echo "== hipBLASLt kernels behind the yardstick" | tee -a $OUT
mkdir -p gpurun_out/prof_c; P=$PWD/gpurun_out/prof_c
( cd /tmp && rocprofv3 --kernel-trace --stats -d $P -o yard -- python $GRAFT_REPO_ROOT/tools/blaslt_yardstick.py > $P/yard.log 2>&1 )
python - <<'PY' | tee -a $OUT
import sqlite3, glob
for db in glob.glob('gpurun_out/prof_c/**/*.db', recursive=True):
    c = sqlite3.connect(db)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if t.startswith('rocpd_kernel_dispatch')][0]
    ks = [t for t in tabs if t.startswith('rocpd_info_kernel_symbol')][0]
    q = f"select s.kernel_name, count(*), avg(d.end-d.start)/1e3, d.workgroup_size_x, d.grid_size_x, d.grid_size_y, d.grid_size_z, d.group_segment_size, s.arch_vgpr_count, s.accum_vgpr_count, s.sgpr_count from {kd} d join {ks} s on d.kernel_id = s.id group by s.kernel_name, d.grid_size_x order by sum(d.end-d.start) desc limit 16"
    for r in c.execute(q): print(r)
PY
rm -rf gpurun_out/prof_c
