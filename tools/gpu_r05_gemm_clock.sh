# gpurun -- 'bash tools/gpu_r05_gemm_clock.sh': encoder GEMM phases + effective shader clock, random vs all-zero operands (profiles/r05_f_gemm_clock_ab.txt)
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
hipcc --offload-arch=gfx950 -O3 -std=c++17 -x hip tools/gemm_bench.cpp speaksense_amd/csrc/kernels_gemm.hip -Ispeaksense_amd/csrc -o /tmp/gemm_bench || exit 1
OUT=gpurun_out/r05_f_gemm_clock_ab.txt
( echo "== random operands (as the engine runs them)"; SS_TRACE=1 /tmp/gemm_bench | grep -v "check" | head -40
  echo; echo "== all-zero operands (same instruction stream, fewer toggling bits)"; SS_GEMM_ZERO=1 SS_TRACE=1 /tmp/gemm_bench | grep -v "check" | head -40
  echo; echo "== random again (drift check)"; /tmp/gemm_bench | grep -v "check" | head -14 ) > $OUT 2>&1
cat $OUT | cut -c1-220
