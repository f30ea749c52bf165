"""Dev tool: per-stage GPU-vs-oracle errors for a preset (python tools/stage_check.py tiny.en)."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from speaksense_amd import ggml_io, synth, binding
from oracle import binding as orc
name = sys.argv[1] if len(sys.argv) > 1 else "tiny.en"
path = f"/tmp/stage_{name}.bin"
if not os.path.exists(path): ggml_io.write_model(path, name, seed=1)
om = orc.OracleModel(path)
eng = binding.Engine(path, dtype=binding.DTYPE_F16, max_batch=1)
pcm = synth.speech_like(4)
mel = om.log_mel(pcm)
print("mel err", np.abs(eng.log_mel(pcm) - mel).max())
ref = om.encode(mel, 0, orc.MODE_GGML_F16); got = eng.encode(mel, 0)
print("encoder rel err", np.abs(got - ref).max() / np.abs(ref).max(), "shape", ref.shape)
enc = om.encode(mel, 0, orc.MODE_F32)
ost = om.new_state(orc.MODE_GGML_F16); ost.set_encoder(enc)
ses = eng.new_session(); ses.set_encoder(enc)
toks = [om.sot, om.beg + 3, 1234, 777, 42, om.beg + 50, 99, 100, 101]
r = ost.decode(toks[:1], 0); g = ses.decode(toks[:1], 0)
print("logits step0 err/std", np.abs(g - r).max() / r.std(), "argmax", g.argmax(), r.argmax())
for i in range(1, len(toks)):
    r = ost.decode(toks[i:i + 1], i); g = ses.decode(toks[i:i + 1], i)
    top2 = np.sort(r)[-2:]
    print(f"step {i}: err/std {np.abs(g - r).max() / r.std():.2e} argmax {g.argmax()} {r.argmax()} top-2 gap/std {(top2[1]-top2[0])/r.std():.2e}")
# the failing case: first decode steps of the real path
ref = om.new_state(orc.MODE_GGML_F16).full(pcm, orc.default_params(language="en", temperature_inc=0.0))
got = eng.new_session().transcribe(pcm, binding.default_params(language="en", temperature_inc=0.0))
print("ref tokens", list(ref["tokens"][:8])); print("got tokens", list(got["tokens"][:8]))
