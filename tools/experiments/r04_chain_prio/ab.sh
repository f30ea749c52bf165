# A/B of wave issue priorities.  Run r04_l: s_setprio 3 in the chain kernels (dec_gemv / dec_reduce_ln / dec_self_attn) against the default priority ->
# +1.4 %, now the product (profiles/r04_l_chain_prio_ab.txt).  This script compares the product library with gpurun_ab/libprio.so, built on the
# build box by build_prio.sh with an extra -D (e.g. -DSS_CROSS_PRIO=1: the cross-attention one level above the encoder GEMMs), on the GPU box:
#   bash tools/experiments/r04_chain_prio/ab.sh <out.jsonl>
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=${1:-gpurun_out/r04_chain_prio_ab.jsonl}; : > $OUT
cp speaksense_amd/libspeaksense_hip.so /tmp/lib_base.so
run() {  # <label> <lib> <lanes> <device-batch> <inflight> <steps>
  cp $2 speaksense_amd/libspeaksense_hip.so
  line=$(python bench.py --no-cpu-baseline --headline-only --steps $6 --warmup $5 --lanes $3 --device-batch $4 --inflight $5 2>/tmp/ab.err)
  if [ $? -ne 0 ]; then echo "{\"label\": \"$1\", \"failed\": true}" >> $OUT; tail -3 /tmp/ab.err; else echo "{\"label\": \"$1\", \"bench\": $line}" >> $OUT; fi
}
for rep in 1 2; do
  run "3 lanes x 32, chain kernels s_setprio 3 (product)" /tmp/lib_base.so 3 32 12 36
  run "3 lanes x 32, + cross-attention s_setprio 1" gpurun_ab/libprio.so 3 32 12 36
done
run "1 lane x 32, chain kernels s_setprio 3 (product)" /tmp/lib_base.so 1 32 4 16
run "1 lane x 32, + cross-attention s_setprio 1" gpurun_ab/libprio.so 1 32 4 16
cp /tmp/lib_base.so speaksense_amd/libspeaksense_hip.so
python tools/experiments/r04_cumask_lanes/summarize.py $OUT
