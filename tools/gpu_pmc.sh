# HBM traffic counters for the dominant kernel: separate --pmc passes (kernel-trace only), per MI355X_MICROARCH.md §HBM
mkdir -p gpurun_out/pmc
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/pmc
for C in FETCH_SIZE WRITE_SIZE; do
  cd /tmp && rocprofv3 --kernel-trace --pmc $C -d $OUT -o pmc_$C -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --fixed-steps 4 > $OUT/bench_$C.log 2> $OUT/bench_$C.err
  cd $GRAFT_REPO_ROOT
done
ls gpurun_out/pmc
python - <<'PY'
import sqlite3, glob
for f in sorted(glob.glob("gpurun_out/pmc/*_results.db")):
    db = sqlite3.connect(f)
    tabs = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
    print(f, [t for t in tabs if 'pmc' in t.lower() or 'counter' in t.lower()][:8])
    try:
        cols = [r[1] for r in db.execute("pragma table_info(counters_collection)")]
        print(cols)
        q = "select kernel_name, grid_size_x, counter_name, count(*), avg(value), min(value), max(value) from counters_collection where kernel_name like '%gemm%' or kernel_name like '%attn%' group by kernel_name, grid_size_x, counter_name order by 5 desc limit 16"
        lines = ["| kernel | grid_x | counter | launches | avg KiB | min | max |", "|---|---|---|---|---|---|---|"]
        for r in db.execute(q): lines.append("| `%s` | %s | %s | %s | %.1f | %.1f | %.1f |" % (str(r[0])[:64], r[1], r[2], r[3], r[4], r[5], r[6]))
        print("\n".join(lines))
        open(f.replace("_results.db", "_summary.md"), "w").write("\n".join(lines) + "\n")
    except Exception as e:
        print("ERR", e)
PY
rm -f gpurun_out/pmc/*.db
