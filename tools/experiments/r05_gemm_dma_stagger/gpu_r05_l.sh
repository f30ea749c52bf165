# gpurun -- 'bash tools/gpu_r05_l.sh': gemm256k64 main loop with the two waves of a SIMD issuing their LDS-DMA in different quarters (-DSS_K64_DMA_STAGGER), A / B / A / B
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
B="hipcc --offload-arch=gfx950 -O3 -std=c++17 -x hip tools/gemm_bench.cpp speaksense_amd/csrc/kernels_gemm.hip -Ispeaksense_amd/csrc"
$B -o /tmp/gb_base 2>/dev/null || exit 1
$B -DSS_K64_DMA_STAGGER -o /tmp/gb_stag 2>/dev/null || exit 1
OUT=gpurun_out/r05_l_gemm_dma_stagger_ab.txt
( for r in 1 2; do
    echo "== baseline (both waves of a SIMD issue 2 DMA per quarter), run $r"; SS_TRACE=1 /tmp/gb_base | grep -v "check\|wg " | head -22
    echo "== staggered (waves 0-3: 4 + 4 in quarters 0, 1; waves 4-7: quarters 2, 3), run $r"; SS_TRACE=1 /tmp/gb_stag | grep -v "wg " | head -30
  done ) > $OUT 2>&1
cut -c1-200 $OUT
