# One parameterised runner for everything that goes to the GPU box (replaces the per-run tools/gpu_*.sh scripts of rounds 1-2).
#   gpurun --timeout N -- 'bash tools/gpu.sh <tag> <recipe> [<recipe> ...]'
# recipes (run in the order given; every artefact lands in gpurun_out/ prefixed with <tag>):
#   tests[:<pytest args>]   pytest -m gpu (default: the whole suite); e.g.  tests:tests/test_gpu_bench_config.py
#   smoke                   __graft_entry__.smoke()
#   bench[:<bench args>]    python bench.py <args>            -> bench_<tag>_<n>.json   (args separated by ',')
#   prof[:<bench args>]     rocprofv3 --kernel-trace --stats of the bench command -> <tag>_<n>_kernel_stats.md
#   pmc[:<dtype>]           separate FETCH_SIZE / WRITE_SIZE --pmc passes of the bench command -> <tag>_pmc.md (+ pmc_traffic.json)
#   py:<script>[,args]      python <script> args              -> <tag>_<n>.log
#   sh:<command>            bash -c <command> ('+' stands for a space)
TAG=$1; shift
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export OMP_WAIT_POLICY=passive TMPDIR=/tmp
n=0
for R in "$@"; do
  n=$((n + 1))
  KIND=${R%%:*}; ARG=""; [ "$R" != "$KIND" ] && ARG=${R#*:}
  ARGS=$(echo "$ARG" | tr ',' ' ')
  case $KIND in
    tests)
      rm -f gpurun_out/parity_report.txt
      ( time timeout ${TEST_TIMEOUT:-3000} python -m pytest ${ARGS:-tests} -q -m gpu ${PYTEST_X--x} --durations=15 ) > gpurun_out/${TAG}_pytest_$n.log 2>&1; echo "pytest rc=$?" >> gpurun_out/${TAG}_pytest_$n.log
      tail -30 gpurun_out/${TAG}_pytest_$n.log | cut -c1-300
      [ -f gpurun_out/parity_report.txt ] && cp gpurun_out/parity_report.txt gpurun_out/${TAG}_parity_report_$n.txt ;;
    smoke)
      timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 ;;
    bench)
      timeout 1200 python bench.py $ARGS > gpurun_out/bench_${TAG}_$n.json 2> gpurun_out/bench_${TAG}_$n.err; echo "bench [$ARGS] rc=$?"
      cut -c1-700 gpurun_out/bench_${TAG}_$n.json; tail -3 gpurun_out/bench_${TAG}_$n.err ;;
    prof)
      mkdir -p gpurun_out/prof; OUT=$PWD/gpurun_out/prof
      ( cd /tmp && rocprofv3 --kernel-trace --stats -d $OUT -o ${TAG}_$n -- python $GRAFT_REPO_ROOT/bench.py --steps ${PROF_STEPS:-8} --warmup ${PROF_WARMUP:-8} --no-cpu-baseline $ARGS > $OUT/bench_${TAG}_$n.log 2> $OUT/bench_${TAG}_$n.err )
      python tools/rocpd_stats.py gpurun_out/prof/${TAG}_${n}_results.db gpurun_out/${TAG}_${n}_kernel_stats.md | cut -c1-200 | head -${PROF_LINES:-45}
      [ -n "$KEEP_DB" ] || rm -f gpurun_out/prof/${TAG}_${n}_results.db ;;
    pmc)
      DT=${ARGS:-f16}; mkdir -p gpurun_out/pmc_$DT; OUT=$PWD/gpurun_out/pmc_$DT
      for C in FETCH_SIZE WRITE_SIZE; do
        ( cd /tmp && rocprofv3 --kernel-trace --pmc $C -d $OUT -o pmc_$C -- python $GRAFT_REPO_ROOT/bench.py --steps 16 --warmup 0 --no-cpu-baseline --headline-only --fixed-steps 6 --dtype $DT > $OUT/bench_$C.log 2> $OUT/bench_$C.err )
      done
      # one traffic file per call of this script: a second pmc recipe (another dtype) adds its entries to the first one's
      [ -f gpurun_out/${TAG}_pmc_traffic.json ] || cp profiles/pmc_traffic.json gpurun_out/${TAG}_pmc_traffic.json 2>/dev/null
      python tools/pmc_summary.py $OUT $TAG gpurun_out/${TAG}_pmc_traffic.json gpurun_out/${TAG}_pmc_$DT.md $DT
      rm -f $OUT/*.db ;;
    py)
      timeout ${PY_TIMEOUT:-1200} python $ARGS > gpurun_out/${TAG}_$n.log 2>&1; echo "py [$ARGS] rc=$?"; tail -${PY_LINES:-40} gpurun_out/${TAG}_$n.log | cut -c1-300 ;;
    sh)
      bash -c "$(echo "$ARG" | tr '+' ' ')" > gpurun_out/${TAG}_$n.log 2>&1; echo "sh rc=$?"; tail -${PY_LINES:-40} gpurun_out/${TAG}_$n.log | cut -c1-300 ;;
    *) echo "unknown recipe $R" ;;
  esac
done
