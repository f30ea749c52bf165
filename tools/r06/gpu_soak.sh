# long soaks on the d = 1280 shape (the driver's run keeps 30 s on the toy model): f16 420 s, fp8 240 s
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp OMP_WAIT_POLICY=passive
rm -f gpurun_out/parity_report.txt
SS_SOAK_SECONDS=420 SS_SOAK_MODEL=wide2 timeout 1500 python -m pytest tests/test_gpu_lifetime.py::test_soak_random_interleavings -q -m gpu 2>&1 | tail -4
SS_SOAK_SECONDS=240 SS_SOAK_MODEL=wide2 SS_SOAK_DTYPE=fp8 timeout 1200 python -m pytest tests/test_gpu_lifetime.py::test_soak_random_interleavings -q -m gpu 2>&1 | tail -4
cp gpurun_out/parity_report.txt gpurun_out/r06_p_soak_report.txt
cat gpurun_out/r06_p_soak_report.txt
