export TMPDIR=/tmp
OUT=$PWD/gpurun_out/gpmc; mkdir -p $OUT
BIN=$PWD/tools/gemm_bench.bin
cd /tmp
for C in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VMEM_RD SQ_INSTS_SALU"; do
  tag=$(echo $C | tr ' ' '_' | cut -c1-40)
  rocprofv3 --kernel-trace --pmc $C -d $OUT -o p_$tag -- $BIN > /dev/null 2> $OUT/err_$tag.txt
done
cd - > /dev/null
python - <<'PY'
import sqlite3, glob
for f in sorted(glob.glob("gpurun_out/gpmc/*_results.db")):
    db = sqlite3.connect(f)
    q = "select kernel_name, grid_size_x, counter_name, avg(value) from counters_collection where kernel_name like '%gemm256%Li1E%' group by kernel_name, grid_size_x, counter_name order by 2,3"
    for r in db.execute(q): print(r[1], r[2], "%.4g" % r[3])
PY
rm -f gpurun_out/gpmc/*.db
