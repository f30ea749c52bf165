cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/prof; export TMPDIR=/tmp; OUT=$PWD/gpurun_out/prof
for D in f16; do
  ( cd /tmp && rocprofv3 --kernel-trace -d $OUT -o tl_$D -- python $GRAFT_REPO_ROOT/bench.py --steps 48 --warmup 12 --no-cpu-baseline --dtype $D > $OUT/tl_$D.log 2> $OUT/tl_$D.err )
  echo "== $D"; python tools/rocpd_timeline.py $OUT/tl_${D}_results.db gpurun_out/r02_z_timeline_$D.md | cut -c1-400
  rm -f $OUT/tl_${D}_results.db
done
