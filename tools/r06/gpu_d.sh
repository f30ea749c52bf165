# Round 6, GPU call D: why did the 2-rank bench test take 79 s; tiny-context test; gemm_bench sustained-load check
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp OMP_WAIT_POLICY=passive
OUT=gpurun_out/r06_d.txt; : > $OUT
python -c "
from speaksense_amd import ggml_io; import bench, os
p = bench.model_path_for('base.en')
if not os.path.exists(p): ggml_io.write_model(p, 'base.en', seed=0)"
for aff in "" "--no-affinity"; do
  /usr/bin/time -f "2 ranks [$aff]: %e s wall, %U user, %S sys" env SS_BENCH_DEVICE=0 MASTER_ADDR=127.0.0.1 python bench.py --gpus 2 --model base.en --steps 3 --warmup 1 --inflight 2 --lanes 2 --dist-backend gloo --no-cpu-baseline --no-steady $aff 2>&1 | grep -E 'ranks|^\{' | cut -c1-300 | tee -a $OUT
done
/usr/bin/time -f "8 ranks: %e s wall, %U user, %S sys" env SS_BENCH_DEVICE=0 MASTER_ADDR=127.0.0.1 python bench.py --gpus 8 --model base.en --steps 3 --warmup 1 --inflight 2 --lanes 2 --device-batch 16 --dist-backend gloo --no-cpu-baseline --no-steady --headline-only 2>&1 | grep -E 'ranks|^\{' | cut -c1-1500 | tee -a $OUT
( timeout 900 python -m pytest tests/test_gpu_audio_ctx.py -q -m gpu -k "tiny or mixed_context_batch" 2>&1 | tail -8 ) | tee -a $OUT
echo "== gemm_bench: is M = 12000 also slower when the launches run for longer?  (SS_GEMM_REPS)" | tee -a $OUT
for R in 10 60; do echo "-- reps $R" | tee -a $OUT; SS_GEMM_REPS=$R ./tools/gemm_bench.bin 2>&1 | grep -E 'store|gelu' | grep -E '^(FC1|FC1x4|QK|QKx4) ' | tee -a $OUT; done
