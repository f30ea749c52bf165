# Round 6, GPU call E: 2-rank / 8-rank bench wall time; vendor yardstick at equal sustained load (60 launches each), interleaved A/B/A
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp OMP_WAIT_POLICY=passive
OUT=gpurun_out/r06_e.txt; : > $OUT
python -c "
from speaksense_amd import ggml_io; import bench, os
p = bench.model_path_for('base.en')
if not os.path.exists(p): ggml_io.write_model(p, 'base.en', seed=0)"
for aff in "" "--no-affinity"; do
  t0=$(date +%s.%N)
  SS_BENCH_DEVICE=0 MASTER_ADDR=127.0.0.1 python bench.py --gpus 2 --model base.en --steps 3 --warmup 1 --inflight 2 --lanes 2 --dist-backend gloo --no-cpu-baseline --no-steady $aff > /tmp/b2.json 2>/tmp/b2.err
  t1=$(date +%s.%N)
  echo "2 ranks [$aff]: $(python -c "print(round($t1 - $t0, 1))") s wall; $(python -c "
import json; d = json.loads(open('/tmp/b2.json').read().strip().splitlines()[-1]); print(d['value'], d['host_cost'])")" | tee -a $OUT
  tail -2 /tmp/b2.err | cut -c1-200 | tee -a $OUT
done
t0=$(date +%s.%N)
SS_BENCH_DEVICE=0 MASTER_ADDR=127.0.0.1 python bench.py --gpus 8 --model base.en --steps 3 --warmup 1 --inflight 2 --lanes 2 --device-batch 16 --dist-backend gloo --no-cpu-baseline --no-steady --headline-only > /tmp/b8.json 2>/tmp/b8.err
t1=$(date +%s.%N)
echo "8 ranks: $(python -c "print(round($t1 - $t0, 1))") s wall; $(python -c "
import json; d = json.loads(open('/tmp/b8.json').read().strip().splitlines()[-1]); print(d['value'], d['n_gpus'], d['config']['chunks_per_step'], d['host_cost'])")" | tee -a $OUT
echo "== yardstick at equal sustained load: 60 launches per line, ours / vendor / ours" | tee -a $OUT
SS_GEMM_REPS=60 ./tools/gemm_bench.bin 2>&1 | grep -E ' store ' | grep -E '^(FC1|FC2|QKV|QK|O|crossKV|FC1x4|FC2x4|Ox4|QKx4) ' | tee -a $OUT
SS_YARD_REPS=60 python tools/blaslt_yardstick.py 2>&1 | grep hipBLASLt | tee -a $OUT
SS_GEMM_REPS=60 ./tools/gemm_bench.bin 2>&1 | grep -E ' store ' | grep -E '^(FC1|FC2|QKV|QK|O|crossKV|FC1x4|FC2x4|Ox4|QKx4) ' | tee -a $OUT
