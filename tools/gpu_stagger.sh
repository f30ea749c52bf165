mkdir -p gpurun_out
for cfg in "0 1" "3000 4" "6000 4" "9000 4" "3000 8" "6000 8" "12000 4" "6000 2" "12000 2"; do set -- $cfg
  echo "== stagger $1 ticks x $2 groups"; SS_GEMM_STAGGER=$1 SS_GEMM_STAGGER_GROUPS=$2 timeout 120 ./tools/gemm_bench.bin 2>&1 | grep -E "^(FC1|FC2|QKV|O ) " 
done | tee gpurun_out/stagger.log
