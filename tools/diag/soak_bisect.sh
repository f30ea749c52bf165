# the soak test (90 s, 3 runs) from the worktrees under .bisect/: which commit starts to corrupt the host heap?  (dev tool, GPU box)
cd $GRAFT_REPO_ROOT
export SS_SOAK_SECONDS=${SS_SOAK_SECONDS:-90}
for c in $BISECT_COMMITS; do
  for i in 1 2 3; do
    ( cd .bisect/$c && python -m pytest tests/test_gpu_lifetime.py -q -m gpu -k soak -p no:faulthandler > ../../gpurun_out/bisect_${c}_$i.log 2>&1 ); rc=$?
    echo "$c run $i rc=$rc $(tail -1 gpurun_out/bisect_${c}_$i.log | cut -c1-100)"
  done
done
