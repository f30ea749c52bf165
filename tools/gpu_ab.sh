# A/B a decode-path change: parity subset + bench with and without an env setting ($1, e.g. SS_DECODE_GRAPH=0)
mkdir -p gpurun_out
export OMP_WAIT_POLICY=passive
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_variants.py -q -x > gpurun_out/pytest_ab.log 2>&1; tail -3 gpurun_out/pytest_ab.log
timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/bench_new.log 2>gpurun_out/bench_new.err; cut -c1-330 gpurun_out/bench_new.log; tail -2 gpurun_out/bench_new.err
if [ -n "$1" ]; then env $1 timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/bench_old.log 2>/dev/null; cut -c1-330 gpurun_out/bench_old.log; fi
