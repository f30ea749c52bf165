# final record of round 6 on the committed tree: the driver's three steps (GPU tests, smoke, bench with its own flags)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp OMP_WAIT_POLICY=passive
rm -f gpurun_out/parity_report.txt
( time timeout 2400 python -m pytest tests -x -q -m gpu ) > gpurun_out/r06_last_pytest.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/r06_last_pytest.log
cp gpurun_out/parity_report.txt gpurun_out/r06_last_parity_report.txt 2>/dev/null
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_r06_last_driver_command.json 2> gpurun_out/bench_r06_last.err; echo "bench rc=$?"
python -c "
import json
d = json.loads(open('gpurun_out/bench_r06_last_driver_command.json').read().strip().splitlines()[-1]); r = d['roofline']
print(d['value'], d['ms_per_step'], d['config']['workload'][:120]); print('frac', r['frac'], 'pass', r['avg_launch_ms'], 'traffic/alg', r['traffic'] / r['algorithmic_bytes'] if r['traffic'] else None, 'mfma', r['mfma_bound_half']['frac'], 'cpu', d['cpu_baseline']['value'], d['cpu_baseline']['cores'])"
