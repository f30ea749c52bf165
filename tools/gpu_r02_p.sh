cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
for ko in 0 1 2 3 4 7 8 15 16 20 28 0; do SS_F8_KO=$ko timeout 120 python tools/gemm_fp8_ko.py 2>&1 | tail -1; done | tee gpurun_out/r02_p_gemm_fp8_knockouts.txt
