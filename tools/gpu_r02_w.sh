# round 2, run O: full GPU suite at the fp8 commit, smoke, default + fp8 bench lines (with cpu_baseline on the default), rocprofv3 kernel stats of both
TAG=${1:-r02_w}
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export OMP_WAIT_POLICY=passive
rm -f gpurun_out/parity_report.txt
( time timeout 2400 python -m pytest tests -q -m gpu ) > gpurun_out/pytest_gpu_$TAG.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu_$TAG.log
tail -12 gpurun_out/pytest_gpu_$TAG.log | cut -c1-250
cp gpurun_out/parity_report.txt gpurun_out/${TAG}_parity_report.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 900 python bench.py > gpurun_out/bench_${TAG}_default_f16.json 2> gpurun_out/bench_${TAG}_default_f16.err; echo "bench default rc=$?"; cut -c1-400 gpurun_out/bench_${TAG}_default_f16.json
timeout 900 python bench.py --dtype fp8 --no-cpu-baseline > gpurun_out/bench_${TAG}_fp8.json 2> gpurun_out/bench_${TAG}_fp8.err; echo "bench fp8 rc=$?"; cut -c1-400 gpurun_out/bench_${TAG}_fp8.json
PROF_STEPS=8 PROF_WARMUP=8 bash tools/gpu_prof.sh ${TAG}_f16 | head -40
PROF_STEPS=8 PROF_WARMUP=8 PROF_ARGS="--dtype fp8" bash tools/gpu_prof.sh ${TAG}_fp8 | head -40
