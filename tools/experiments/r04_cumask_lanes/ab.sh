# A/B: lanes sharing the chip (default) vs lanes confined to disjoint CU slices (SS_LANE_CUS=1, engine.cpp create_lane_stream).
# Every point keeps ~96 chunks in flight (lanes x rows per pass) unless it says otherwise; headline only (timed region), Mode F, f16.
#   bash tools/experiments/r04_cumask_lanes/ab.sh <out file>      (on the GPU box; summary: python tools/experiments/r04_cumask_lanes/summarize.py <out file>)
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=${1:-gpurun_out/r04_cumask_ab.jsonl}; : > $OUT
run() {  # <label> <env> <lanes> <device-batch> <inflight>
  line=$(env $2 python bench.py --no-cpu-baseline --headline-only --steps 36 --warmup 12 --lanes $3 --device-batch $4 --inflight $5 2>/tmp/ab.err)
  if [ $? -ne 0 ]; then echo "{\"label\": \"$1\", \"failed\": true}" >> $OUT; tail -3 /tmp/ab.err; else echo "{\"label\": \"$1\", \"bench\": $line}" >> $OUT; fi
}
run "shared  3 lanes x 32" "SS_X=0" 3 32 12
run "sliced  3 lanes x 32" "SS_LANE_CUS=1" 3 32 12
run "shared  4 lanes x 24" "SS_X=0" 4 24 12
run "sliced  4 lanes x 24" "SS_LANE_CUS=1" 4 24 12
run "shared  2 lanes x 48" "SS_X=0" 2 48 12
run "sliced  2 lanes x 48" "SS_LANE_CUS=1" 2 48 12
run "sliced  8 lanes x 12" "SS_LANE_CUS=1" 8 12 12
run "sliced  4 lanes x 32 (128 in flight)" "SS_LANE_CUS=1" 4 32 16
run "shared  4 lanes x 32 (128 in flight)" "SS_X=0" 4 32 16
run "sliced  3 lanes x 32, no graphs" "SS_LANE_CUS=1 SS_DECODE_GRAPH=0" 3 32 12
python tools/experiments/r04_cumask_lanes/summarize.py $OUT
