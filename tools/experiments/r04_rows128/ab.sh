# A/B: rows per decoder pass beyond 64 (round 4: kPartRows = 128, CT = 8 column tiles in dec_gemv_kernel) against the 3 x 32 default.
#   bash tools/experiments/r04_rows128/ab.sh <out.jsonl>   (GPU box; summary printed at the end)
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=${1:-gpurun_out/r04_rows128_ab.jsonl}; : > $OUT
run() {  # <label> <lanes> <device-batch> <inflight> <steps>
  line=$(python bench.py --no-cpu-baseline --headline-only --steps $5 --warmup $4 --lanes $2 --device-batch $3 --inflight $4 2>/tmp/ab.err)
  if [ $? -ne 0 ]; then echo "{\"label\": \"$1\", \"failed\": true}" >> $OUT; tail -3 /tmp/ab.err; else echo "{\"label\": \"$1\", \"bench\": $line}" >> $OUT; fi
}
run "3 lanes x  32 rows,  96 in flight" 3 32 12 36
run "2 lanes x  64 rows, 128 in flight" 2 64 16 48
run "1 lane  x 128 rows, 128 in flight" 1 128 16 48
run "1 lane  x  96 rows,  96 in flight" 1 96 12 36
run "2 lanes x  96 rows, 192 in flight" 2 96 24 72
run "2 lanes x 128 rows, 256 in flight" 2 128 32 96
run "3 lanes x  64 rows, 192 in flight" 3 64 24 72
python tools/experiments/r04_cumask_lanes/summarize.py $OUT
