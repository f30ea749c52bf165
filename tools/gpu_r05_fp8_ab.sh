# gpurun -- 'bash tools/gpu_r05_fp8_ab.sh': the e4m3 GEMM with 128-byte LDS rows (gemm_f8k128_kernel, default) against the 64-byte-row main loop (SS_F8_K128=0),
# A / B / A on one box, then the GEMM self-tests and the fp8 parity tests on the new kernel
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
OUT=gpurun_out/r05_h_gemm_fp8_k128_ab.txt
( echo "== SS_F8_K128=1 (128-byte rows, 2 x 64 KB ring)"; python tools/gemm_fp8_bench.py
  echo "== SS_F8_K128=0 (64-byte rows, 4 x 32 KB ring)"; SS_F8_K128=0 python tools/gemm_fp8_bench.py
  echo "== SS_F8_K128=1 again"; python tools/gemm_fp8_bench.py ) > $OUT 2>&1
cut -c1-200 $OUT | tail -60
bash tools/gpu.sh r05_h tests:tests/test_gpu_gemm.py,tests/test_gpu_fp8.py
