# rocprofv3 kernel trace of the bench command (run on the GPU box via gpurun)
mkdir -p gpurun_out/prof
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/prof
cd /tmp && rocprofv3 --kernel-trace --stats -d $OUT -o r01 -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $OUT/bench_prof.log 2> $OUT/bench_prof.err
cd $GRAFT_REPO_ROOT
ls -R gpurun_out/prof | head -30
python - <<'PY'
import csv, glob, collections
fs = glob.glob("gpurun_out/prof/**/*kernel_stats.csv", recursive=True)
print(fs)
for f in fs:
    rows = list(csv.DictReader(open(f)))
    for r in rows[:40]:
        print({k: r[k] for k in r if k in ("Name","Calls","TotalDurationNs","AverageNs","Percentage")})
PY
