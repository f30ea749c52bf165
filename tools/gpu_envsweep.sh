# bench under several env settings on one box: bash tools/gpu_envsweep.sh "A=1" "B=2" ...
mkdir -p gpurun_out
for cfg in "$@"; do
  echo "== $cfg"; env $cfg timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['phase_ms'], d['phase_roofline']['decode_step_ms'], d['roofline']['achieved'])"
done | tee gpurun_out/envsweep.log
