# bench.py with extra args under several env settings: bash tools/gpu_envsweep2.sh "<bench args>" ENV1=.. ENV2=..
mkdir -p gpurun_out
ARGS="$1"; shift
for cfg in "$@"; do
  echo "== $cfg"; env $cfg timeout 600 python bench.py $ARGS --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['phase_ms'], d['phase_roofline']['decode_step_ms'])"
done | tee gpurun_out/envsweep2.log
