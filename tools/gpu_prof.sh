# rocprofv3 kernel trace of the bench command (run on the GPU box via gpurun); summary printed by tools/rocpd_stats.py
mkdir -p gpurun_out/prof
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/prof
TAG=${1:-r01}
cd /tmp && rocprofv3 --kernel-trace --stats -d $OUT -o $TAG -- python $GRAFT_REPO_ROOT/bench.py --steps ${PROF_STEPS:-2} --warmup ${PROF_WARMUP:-8} --no-cpu-baseline $PROF_ARGS > $OUT/bench_$TAG.log 2> $OUT/bench_$TAG.err
cd $GRAFT_REPO_ROOT
python tools/rocpd_stats.py gpurun_out/prof/${TAG}_results.db gpurun_out/prof/${TAG}_kernel_stats.md | cut -c1-180
rm -f gpurun_out/prof/${TAG}_results.db
