# A/B: in-chain weight prefetch (SS_DEC_PREFETCH=<workgroups>: the three reduce + LayerNorm launches of a decoder layer also read the weight matrix
# of the GEMV that follows them -- QKV, cross-q, FC1 -- on the XCD whose consumer workgroups will want it; engine.cpp / kernels_decode.hip).
#   bash tools/experiments/r04_dec_prefetch/ab.sh <out.jsonl>
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=${1:-gpurun_out/r04_dec_prefetch_ab.jsonl}; : > $OUT
run() {  # <label> <env> <lanes> <device-batch> <inflight> <steps>
  line=$(env $2 python bench.py --no-cpu-baseline --headline-only --steps $6 --warmup $5 --lanes $3 --device-batch $4 --inflight $5 2>/tmp/ab.err)
  if [ $? -ne 0 ]; then echo "{\"label\": \"$1\", \"failed\": true}" >> $OUT; tail -3 /tmp/ab.err; else echo "{\"label\": \"$1\", \"bench\": $line}" >> $OUT; fi
}
for pf in 0 256 512 0 256; do
  run "1 lane  x 32, prefetch $pf" "SS_DEC_PREFETCH=$pf" 1 32 4 16
done
for pf in 0 256 512 0 256; do
  run "3 lanes x 32, prefetch $pf" "SS_DEC_PREFETCH=$pf" 3 32 12 36
done
python tools/experiments/r04_cumask_lanes/summarize.py $OUT
