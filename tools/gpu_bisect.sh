mkdir -p gpurun_out
export OMP_WAIT_POLICY=passive
for cfg in "X=1" "SS_ATTN_LDS=0" "SS_DECODE_GRAPH=0" "X=2"; do
  echo "== $cfg"; env $cfg timeout 600 python -m pytest tests/test_gpu_frontend.py -q -k rest_pipeline_end_to_end 2>&1 | tail -1
done
for cfg in "X=1" "SS_DECODE_GRAPH=0"; do
  echo "== modeN $cfg"; env $cfg timeout 900 python bench.py --steps 2 --warmup 1 --fixed-steps 0 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['value'], j['phase_ms'])"
done
