"""dev diagnostic (GPU box): for which audio_ctx does a chunk decoded with forced fallbacks differ when a second chunk of the SAME context shares its encoder pass?"""
import os, sys, tempfile
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from speaksense_amd import binding, ggml_io, synth
path = os.path.join(tempfile.mkdtemp(), "toy.bin")
ggml_io.write_model(path, "toy", seed=1)
key = lambda r: [int(t) for t in r["trace"]]
for dtype, name in ((binding.DTYPE_F16, "f16"), (binding.DTYPE_BF16, "bf16")):
    eng = binding.Engine(path, dtype=dtype, max_batch=8, n_lanes=1, batch_wait_us=200000)
    for A in (0, 256, 500, 512, 640, 700, 704, 752, 764, 768, 1000, 1024, 1280, 1496):
        out = []
        for seed in (1, 2, 3):
            X = synth.speech_like(seed, 16000 * 3)
            P = binding.default_params(language="en", audio_ctx=A, temperature_inc=0.2, logprob_thold=0.0)
            alone = key(eng.new_session().transcribe(X, P))
            res = []
            for n in (2, 3):
                ss = [eng.new_session() for _ in range(n)]
                ts = [s.submit(X, P) for s in ss]
                res.append(all(key(s.wait(t)) == alone for s, t in zip(ss, ts)))
            out.append(res)
        print(f"{name} audio_ctx {A or 1500}: copies in one encoder pass equal the single run (2 copies, 3 copies) for three audios: {out}", flush=True)
    eng.close()
