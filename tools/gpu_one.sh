mkdir -p gpurun_out
export OMP_WAIT_POLICY=passive
timeout 1200 python -m pytest "$@" > gpurun_out/pytest_one.log 2>&1; echo "rc=$?" >> gpurun_out/pytest_one.log; tail -25 gpurun_out/pytest_one.log
