# gpurun -- 'bash tools/experiments/r05_gemm_mfma32/gpu_r05_t.sh': gemm256k64_kernel with v_mfma_f32_32x32x16 (SS_GEMM_M32=1) against the 16x16x32 form: self-tests, then A / B / A / B
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
OUT=gpurun_out/r05_t_gemm_mfma32_ab.txt
( echo "== self-tests of the 32x32x16 form (tests/test_gpu_gemm.py::test_gemm_matches_reference under SS_GEMM_M32=1)"
  SS_GEMM_M32=1 timeout 600 python -m pytest tests/test_gpu_gemm.py -q -m gpu -k test_gemm_matches_reference -p no:cacheprovider 2>&1 | tail -4
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -x hip tools/gemm_bench.cpp speaksense_amd/csrc/kernels_gemm.hip -Ispeaksense_amd/csrc -o /tmp/gemm_bench 2>/dev/null || exit 1
  for r in 1 2; do
    echo "== 16x16x32 (product), run $r"; SS_TRACE=1 /tmp/gemm_bench | grep -v "wg \|check" | head -24
    echo "== 32x32x16 (SS_GEMM_M32=1), run $r"; SS_GEMM_M32=1 SS_TRACE=1 /tmp/gemm_bench | grep -v "wg " | head -32
  done ) > $OUT 2>&1
cut -c1-220 $OUT
