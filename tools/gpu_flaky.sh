mkdir -p gpurun_out
export OMP_WAIT_POLICY=passive
for cfg in "X=1" "X=2" "SS_DECODE_CHAIN=0" "SS_DECODE_CHAIN=0" "SS_DECODE_GRAPH=0" "X=3"; do
  echo "== $cfg"; env $cfg timeout 600 python -m pytest tests/test_gpu_parity.py -q -k "greedy_identical" 2>&1 | tail -1
done
