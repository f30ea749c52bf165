"""Dev tool: the elements of the fp8 engine's first quantisation point that land on another e4m3 code than the oracle's (see tests/test_gpu_fp8.py)."""
import sys, os, numpy as np, tempfile
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), 'tests'))
from speaksense_amd import binding, synth, ggml_io
from oracle import binding as orc
from test_gpu_fp8 import _e4m3_values
name = sys.argv[1] if len(sys.argv) > 1 else 'toy256'
p = os.path.join(tempfile.mkdtemp(), 't.bin'); ggml_io.write_model(p, name, seed=int(sys.argv[2]) if len(sys.argv) > 2 else 1)
om = orc.OracleModel(p); eng = binding.Engine(p, dtype=binding.DTYPE_FP8, max_batch=1)
mel = om.log_mel(synth.speech_like(5)); tab = _e4m3_values(); pos = np.sort(tab[:127])
codes, exps = eng.fp8_first_quant(mel, 0)
sd = np.repeat(np.exp2(exps.astype(np.float32) - 127), 64, axis=1)
got = tab[codes] * sd; ref = om.encode_fp8_first_quant(mel, 0)
blk = np.repeat(np.abs(ref).reshape(ref.shape[0], -1, 64).max(axis=2), 64, axis=1)
ii = np.nonzero(got != ref)
a = np.abs(ref[ii]) / sd[ii]; b = np.abs(got[ii]) / sd[ii]
ia = np.clip(np.searchsorted(pos, a), 0, 126); ib = np.clip(np.searchsorted(pos, b), 0, 126)
ong = pos[ia] == a
st = np.abs(ia - ib)
print('differ', len(a), 'ref on device grid', ong.mean(), 'steps hist', np.bincount(st[ong])[:8])
one = (pos[np.clip(ia + 1, 0, 126)] - pos[ia]) * sd[ii]
ex = (np.abs(got[ii] - ref[ii]) - one) / blk[ii]
o = np.argsort(-ex)[:12]
for k in o:
    r, c = ii[0][k], ii[1][k]
    print(f"row {r} col {c}: ref {ref[r, c]:.6g} got {got[r, c]:.6g} scale {sd[r, c]:.3g} ref/scale {a[k]:.6g} got/scale {b[k]:.6g} steps {st[k]} on_grid {ong[k]} blkmax/scale {blk[r, c] / sd[r, c]:.5g} excess {ex[k]:.3g}")
