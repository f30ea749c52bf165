"""dev diagnostic (GPU box): chunk X = (speech_like(1, 3 s), audio_ctx 752, forced ladder) alone and submitted together with one companion chunk of each kind."""
import os, sys, tempfile, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from speaksense_amd import binding, ggml_io, synth
path = os.path.join(tempfile.mkdtemp(), "toy.bin")
ggml_io.write_model(path, "toy", seed=1)
eng = binding.Engine(path, max_batch=8, n_lanes=1, batch_wait_us=200000)
X = synth.speech_like(1, 16000 * 3)
PX = dict(language="en", audio_ctx=752, temperature_inc=0.2, logprob_thold=0.0)
key = lambda r: [int(t) for t in r["trace"]]
alone = key(eng.new_session().transcribe(X, binding.default_params(**PX)))
alone2 = key(eng.new_session().transcribe(X, binding.default_params(**PX)))
print("alone twice identical:", alone == alone2, len(alone))
comps = {
    "same chunk": (X, PX),
    "full ctx greedy 3 s": (synth.speech_like(2, 16000 * 3), dict(language="en", temperature_inc=0.0)),
    "full ctx forced ladder 3 s": (synth.speech_like(2, 16000 * 3), dict(language="en", temperature_inc=0.2, logprob_thold=0.0)),
    "ctx 752 greedy 3 s": (synth.speech_like(2, 16000 * 3), dict(language="en", temperature_inc=0.0, audio_ctx=752)),
    "ctx 256 greedy 3 s": (synth.speech_like(2, 16000 * 3), dict(language="en", temperature_inc=0.0, audio_ctx=256)),
    "ctx 752 forced ladder, other audio": (synth.speech_like(2, 16000 * 3), dict(language="en", audio_ctx=752, temperature_inc=0.2, logprob_thold=0.0)),
    "full ctx greedy 44 s": (synth.speech_like(3, 16000 * 44), dict(language="en", temperature_inc=0.0)),
}
for name, (pcm, kw) in comps.items():
    for order in (0, 1):
        s1, s2 = eng.new_session(), eng.new_session()
        if order == 0:
            t1 = s1.submit(X, binding.default_params(**PX)); t2 = s2.submit(pcm, binding.default_params(**kw))
        else:
            t2 = s2.submit(pcm, binding.default_params(**kw)); t1 = s1.submit(X, binding.default_params(**PX))
        r1 = key(s1.wait(t1)); s2.wait(t2)
        i = next((i for i in range(min(len(r1), len(alone))) if r1[i] != alone[i]), None)
        print(f"with [{name}] (X submitted {'first' if order == 0 else 'second'}): {'identical' if r1 == alone else f'DIFFERS from index {i} of {len(alone)}'}", flush=True)
for n in (2, 3, 5):
    ss = [eng.new_session() for _ in range(n)]
    ts = [s.submit(X, binding.default_params(**PX)) for s in ss]
    rs = [key(s.wait(t)) for s, t in zip(ss, ts)]
    print(f"{n} copies of X together: equal to alone {[r == alone for r in rs]}")
eng.close()
