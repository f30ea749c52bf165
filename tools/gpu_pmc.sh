# HBM traffic counters of the bench command: separate --pmc passes (kernel-trace only), per MI355X_MICROARCH.md §HBM.
# usage: [PMC_DTYPE=f16|bf16|fp8] bash tools/gpu_pmc.sh <tag>
TAG=${1:-r02}
mkdir -p gpurun_out/pmc
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/pmc
for C in FETCH_SIZE WRITE_SIZE; do
  cd /tmp && rocprofv3 --kernel-trace --pmc $C -d $OUT -o pmc_$C -- python $GRAFT_REPO_ROOT/bench.py --steps 16 --warmup 0 --no-cpu-baseline --fixed-steps 6 --dtype ${PMC_DTYPE:-f16} > $OUT/bench_$C.log 2> $OUT/bench_$C.err
  cd $GRAFT_REPO_ROOT
done
ls gpurun_out/pmc
cp profiles/pmc_traffic.json gpurun_out/pmc/pmc_traffic.json 2>/dev/null
python tools/pmc_summary.py gpurun_out/pmc $TAG gpurun_out/pmc/pmc_traffic.json gpurun_out/pmc/${TAG}_pmc.md ${PMC_DTYPE:-f16}
rm -f gpurun_out/pmc/*.db
