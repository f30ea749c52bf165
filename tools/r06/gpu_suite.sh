# the driver's GPU step as the driver runs it (-x), with every duration
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp OMP_WAIT_POLICY=passive
rm -f gpurun_out/parity_report.txt
( time timeout 2400 python -m pytest tests -q -m gpu -x --durations=0 ) > gpurun_out/r06_suite_pytest.log 2>&1; echo "pytest rc=$?"
tail -5 gpurun_out/r06_suite_pytest.log
grep -E '^[0-9.]+s (call|setup|teardown)' gpurun_out/r06_suite_pytest.log | awk '{t=$1; sub("s","",t); split($3,a,"::"); f[a[1]]+=t; tot+=t} END{for(k in f) printf "%8.1f %s\n", f[k], k; printf "%8.1f TOTAL\n", tot}' | sort -rn | head -20
cp gpurun_out/parity_report.txt gpurun_out/r06_suite_parity_report.txt 2>/dev/null
