cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py tests/test_gpu_large_v3.py tests/test_gpu_fp8.py -q -m gpu -x -k "encoder or stage or large_v3_full_depth or golden or fp8" 2>&1 | tail -5
for r in 1 2; do
  for v in 0 1; do
    SS_VT_GEMM=$v python bench.py --steps 20 --warmup 4 --no-cpu-baseline --headline-only > gpurun_out/ab_vt_${v}_$r.json 2>/dev/null
    python -c "import json;d=json.loads(open('gpurun_out/ab_vt_${v}_$r.json').read().strip().splitlines()[-1]);print('SS_VT_GEMM=$v run $r:', d['value'], d['ms_per_step'], d['phase_ms'])"
  done
done
