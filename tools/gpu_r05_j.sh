# gpurun -- 'bash tools/gpu_r05_j.sh': V^T epilogue of the encoder's V projection beside the plain-store kinds (tools/gemm_bench), then a kernel trace of the headline region only
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
hipcc --offload-arch=gfx950 -O3 -std=c++17 -x hip tools/gemm_bench.cpp speaksense_amd/csrc/kernels_gemm.hip -Ispeaksense_amd/csrc -o /tmp/gemm_bench 2>/dev/null || exit 1
/tmp/gemm_bench | grep -v check | grep "Ox4\|^O \|QKx4\|FC1x4" > gpurun_out/${TAG:-r05_j}_gemm_vt.txt 2>&1
cat gpurun_out/${TAG:-r05_j}_gemm_vt.txt
PROF_STEPS=16 PROF_WARMUP=8 bash tools/gpu.sh ${TAG:-r05_j} prof:--headline-only,--no-mode-n
