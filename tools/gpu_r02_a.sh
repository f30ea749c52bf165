# round 2, run A: the full GPU suite (incl. the large-v3 full-depth tests), the default bench line, a kernel trace.   usage: bash tools/gpu_r02_a.sh <tag>
TAG=${1:-r02_a}
mkdir -p gpurun_out
export OMP_WAIT_POLICY=passive
nproc > gpurun_out/nproc_$TAG.txt
( time timeout 2400 python -m pytest tests -q -m gpu --durations=25 ) > gpurun_out/pytest_gpu_$TAG.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu_$TAG.log
tail -45 gpurun_out/pytest_gpu_$TAG.log
timeout 900 python bench.py --steps 5 --warmup 2 > gpurun_out/bench_${TAG}_f16.json 2> gpurun_out/bench_${TAG}_f16.err; echo "bench rc=$?"; cut -c1-1500 gpurun_out/bench_${TAG}_f16.json; tail -3 gpurun_out/bench_${TAG}_f16.err
bash tools/gpu_prof.sh ${TAG} > gpurun_out/prof_${TAG}.txt 2>&1; tail -45 gpurun_out/prof_${TAG}.txt | cut -c1-200
