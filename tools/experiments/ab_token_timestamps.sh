cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python -m pytest tests/test_gpu_token_times.py -q -m gpu -x 2>&1 | tail -3
for rep in 1 2; do for flag in "" "--no-token-timestamps"; do
  python bench.py --steps 20 --warmup 8 --no-cpu-baseline --no-mode-n --headline-only $flag 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('token_timestamps %-3s rep $rep: %.1f xRT, %.2f ms per step, mel phase %.2f ms' % ('off' if '$flag' else 'on', d['value'], d['ms_per_step'], d['phase_ms']['mel']))"
done; done
