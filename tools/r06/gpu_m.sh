cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
./tools/diag/store_rate_bench.bin 2>&1 | tee gpurun_out/r06_m_store_rate_bench.txt
