cd "$GRAFT_REPO_ROOT" || exit 1
PMC_DTYPE=f16 bash tools/gpu_pmc.sh r02_u_f16 | tail -32
cp gpurun_out/pmc/pmc_traffic.json profiles/pmc_traffic.json
PMC_DTYPE=fp8 bash tools/gpu_pmc.sh r02_u_fp8 | tail -32
cp gpurun_out/pmc/pmc_traffic.json gpurun_out/pmc_traffic_r02_u.json
