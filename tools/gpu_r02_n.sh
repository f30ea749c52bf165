#!/bin/bash
# round 2, run N: fp8 engine -- GEMM self-tests, parity against the FP8-mode oracle, bench line
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_gemm.py -x -q -k fp8 2>&1 | tail -5
timeout 1500 python -m pytest tests/test_gpu_fp8.py -q 2>&1 | tail -40
cp gpurun_out/parity_report.txt gpurun_out/r02_n_parity_report.txt 2>/dev/null
timeout 900 python bench.py --dtype fp8 --no-cpu-baseline > gpurun_out/bench_r02_n_fp8.json 2> gpurun_out/bench_r02_n_fp8.err; tail -3 gpurun_out/bench_r02_n_fp8.err; cut -c1-1500 gpurun_out/bench_r02_n_fp8.json
