"""Quick manual GPU sanity/timing script (not a pytest file)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np
from speaksense_amd import ggml_io, synth, binding
from oracle import binding as orc
os.makedirs("/tmp/m", exist_ok=True)
for name in ("toy.en", "toy"):
    path = f"/tmp/m/{name}.bin"
    ggml_io.write_model(path, name, seed=1)
    om = orc.OracleModel(path)
    for dt, mode in ((binding.DTYPE_F16, orc.MODE_GGML_F16), (binding.DTYPE_BF16, orc.MODE_BF16)):
        eng = binding.Engine(path, dtype=dt, max_batch=4)
        for seed in (3, 4, 5, 6):
            pcm = synth.speech_like(seed)
            t = time.time(); ref = om.new_state(mode).full(pcm, orc.default_params(language="en")); t_o = time.time() - t
            t = time.time(); got = eng.new_session().transcribe(pcm, binding.default_params(language="en")); t_g = time.time() - t
            g, r = list(got["tokens"]), list(ref["tokens"])
            k = next((i for i in range(min(len(g), len(r))) if g[i] != r[i]), None)
            if k is None and len(g) != len(r): k = min(len(g), len(r))
            print(name, "f16" if dt else "bf16", seed, "equal" if k is None else f"DIVERGE@{k}", len(g), len(r), "enc", got["n_encode"], ref["n_encode"],
                  "fail", got["n_fail"], ref["n_fail"], "dec", got["n_decode"], ref["n_decode"], "t_oracle %.2f t_gpu %.3f" % (t_o, t_g), flush=True)
        eng.close()
