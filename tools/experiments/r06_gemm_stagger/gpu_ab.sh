# Round 6, GPU call K: start stagger of half the persistent GEMM workgroups (de-phase the epilogue bursts), gemm_bench A/B at sustained load
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; OUT=gpurun_out/r06_k_gemm_stagger.txt; : > $OUT
for P in 0 50 0 50 25 75; do echo "-- stagger $P % of a tile (x4 shapes: >= 6 rounds)" | tee -a $OUT
  SS_GEMM_STAGGER=$P SS_GEMM_REPS=40 ./tools/gemm_bench_stagger.bin 2>&1 | grep -E ' (store|gelu|res_f32) ' | grep -E '^(FC1x4|FC2x4|Ox4|QKx4) ' | tee -a $OUT; done
echo "-- stagger 50 %, also the 4-round shapes (M = 12000)" | tee -a $OUT
for P in 0 50; do SS_GEMM_STAGGER=$P SS_GEMM_STAGGER_ROUNDS=2 SS_GEMM_REPS=40 ./tools/gemm_bench_stagger.bin 2>&1 | grep -E ' (store|gelu|res_f32) ' | grep -E '^(FC1|FC2|O|QK) ' | sed "s/^/[$P] /" | tee -a $OUT; done
