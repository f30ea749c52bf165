# A/B of two builds of the library on one box, alternating: speaksense_amd/libspeaksense_hip_prev.so (copied aside before the change) vs the tree's build
cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do for which in prev new; do
  if [ $which = prev ]; then export SS_LIB_PATH=$PWD/speaksense_amd/libspeaksense_hip_prev.so; else unset SS_LIB_PATH; fi
  python bench.py --steps 24 --warmup 8 --no-cpu-baseline --no-mode-n --headline-only $AB_ARGS 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
r = d['roofline']
print('$which rep $rep: %.1f xRT, %.2f ms per step, encoder phase %.2f ms, decode phase %.2f ms, decoder pass %.3f ms at %.1f rows, frac %.4f' % (d['value'], d['ms_per_step'], d['phase_ms']['encode_cross_kv'], d['phase_ms']['decode'], r['avg_launch_ms'], r['rows_per_launch'], r['frac']))"
done; done
