#!/bin/bash
# round 2, run M: fp8 GEMM correctness + A/B timing against the f16 kernel
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_gemm.py -x -q -k fp8 2>&1 | tail -15
timeout 600 python tools/gemm_fp8_bench.py 2>&1 | tee gpurun_out/r02_m_gemm_fp8_bench.txt | tail -30
