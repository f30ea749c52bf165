# A/B the in-tree library against gpurun_ab/libvariant.so on the same box: A B A B
mkdir -p gpurun_out
run() { timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['phase_ms'], d['phase_roofline']['decode_step_ms'], d['roofline']['achieved'])"; }
cp speaksense_amd/libspeaksense_hip.so /tmp/libA.so
for i in 1 2; do
  cp /tmp/libA.so speaksense_amd/libspeaksense_hip.so; run A
  cp gpurun_ab/libvariant.so speaksense_amd/libspeaksense_hip.so; run B
done | tee gpurun_out/abso.log
cp /tmp/libA.so speaksense_amd/libspeaksense_hip.so
