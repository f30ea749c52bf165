# the full-length case lists the driver's run shortens (tests/conftest.py SLOW): run by hand, report committed under profiles/
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp OMP_WAIT_POLICY=passive SS_RUN_SLOW=1
rm -f gpurun_out/parity_report.txt
( time timeout 3000 python -m pytest tests/test_gpu_parity.py tests/test_gpu_lifetime.py::test_soak_random_interleavings tests/test_gpu_bench_config.py -q -m gpu --durations=10 ) > gpurun_out/r06_slow_pytest.log 2>&1; echo "pytest rc=$?"
tail -18 gpurun_out/r06_slow_pytest.log | cut -c1-200
cp gpurun_out/parity_report.txt gpurun_out/r06_slow_parity_report.txt 2>/dev/null
