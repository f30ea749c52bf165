"""dev diagnostic (GPU box): forced-replay margins of the fp8 engine against the oracle's FP8 mode for several audio_ctx values on wide2 (d = 1280):
is a shortened context noisier than the full one (a tail-tile problem would show at contexts that are not multiples of 256)?"""
import os, sys, tempfile
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from speaksense_amd import binding, ggml_io, synth
from oracle import binding as orc

tmp = tempfile.mkdtemp()
path = os.path.join(tmp, "wide2.bin")
ggml_io.write_model(path, "wide2", seed=63, **ggml_io.NATURAL)
om = orc.OracleModel(path)
for dtype, omode, name in ((binding.DTYPE_FP8, orc.MODE_FP8, "fp8"), (binding.DTYPE_F16, orc.MODE_GGML_F16, "f16")):
    eng = binding.Engine(path, dtype=dtype, max_batch=2)
    for A in (1500, 1024, 1000, 752, 512, 256):
        gaps_all, n_steps = [], 0
        for seed in (31, 32, 33):
            pcm = synth.speech_like(seed)
            kw = dict(language="en", temperature_inc=0.0, audio_ctx=A)
            got = eng.new_session().transcribe(pcm, binding.default_params(**kw))
            rep = om.new_state(omode).full(pcm, orc.default_params(**kw), forced=got["sampled"])
            g = np.asarray(rep["forced_gap"])
            gaps_all += [float(x) for x in g[g > 0]]
            n_steps += len(g)
        print(f"{name} audio_ctx {A}: {n_steps} greedy steps, {len(gaps_all)} picks differ from the oracle's argmax, gaps {sorted(round(x, 3) for x in gaps_all)[-6:]}", flush=True)
    eng.close()
