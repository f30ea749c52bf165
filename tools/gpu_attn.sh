mkdir -p gpurun_out
export OMP_WAIT_POLICY=passive
./tools/waveops_test.bin | tr '\n' ';'; echo
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "encoder or greedy or ladder or mel" > gpurun_out/pytest_attn.log 2>&1; tail -3 gpurun_out/pytest_attn.log
for v in 4 3; do
  echo "== SS_ATTN_LDS=$v"; SS_ATTN_LDS=$v timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['value'], j['phase_ms'], j['phase_roofline']['decode_step_ms'], j['roofline']['achieved'])"
done | tee gpurun_out/attn.log
