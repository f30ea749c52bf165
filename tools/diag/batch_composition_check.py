"""Does a chunk's result depend on what it is batched with, and if so is the difference a near tie?  (dev tool, GPU box)
   python tools/diag/batch_composition_check.py
The row count of a decoder pass selects kernel variants and split-K plans, i.e. the f32 summation order; a pick whose runner-up is inside the
f16 noise may then differ between compositions.  For the soak test's audios and parameter variants: serial result vs the result inside batches of
2..8 mixed chunks; every difference is replayed on the oracle and its top-2 margin printed."""
import os, sys, tempfile
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from speaksense_amd import binding, ggml_io, synth
from oracle import binding as orc

path = os.path.join(tempfile.mkdtemp(), "toy.bin")
ggml_io.write_model(path, "toy", seed=1)
eng = binding.Engine(path, max_batch=8, n_lanes=3)
om = orc.OracleModel(path)
lengths = [0.5, 1.0, 3.0, 7.5, 12.0, 29.9, 30.1, 44.0]
variants = {"greedy": dict(), "no_ts": dict(no_timestamps=1), "single": dict(single_segment=1), "maxtok": dict(max_tokens=12), "offset": dict(offset_ms=1500),
            "ladder": dict(temperature_inc=0.2)}   # a fallback chunk decodes with best_of 5 rows: passes of up to 40 rows, other kernel variants
audio = {(sd, ln): synth.speech_like(sd, int(16000 * ln)) for sd in (1, 2, 3) for ln in lengths}
def P(v, mod=binding):
    kw = dict(language="en", temperature_inc=0.0); kw.update(variants[v])
    return mod.default_params(**kw)
keys = [(a, v) for a in audio for v in variants]
serial = {}
for k in keys:
    s = eng.new_session(); serial[k] = s.transcribe(audio[k[0]], P(k[1])); s.close()
rng = np.random.default_rng(0)
n_diff = n_run = 0
worst = 0.0
for trial in range(150):
    n = int(rng.integers(2, 9))
    pick = [keys[int(i)] for i in rng.integers(0, len(keys), n)]
    ses = [eng.new_session() for _ in pick]
    tickets = [s.submit(audio[k[0]], P(k[1])) for s, k in zip(ses, pick)]
    for s, t, k in zip(ses, tickets, pick):
        got = s.wait(t); s.close(); n_run += 1
        if k[1] != "ladder" and list(got["tokens"]) != list(serial[k]["tokens"]):
            n_diff += 1
            margins = []
            for res in (got, serial[k]):
                rep = om.new_state(orc.MODE_GGML_F16).full(audio[k[0]], P(k[1], orc), forced=res["sampled"])
                margins.append(float(rep["forced_gap"].max()) if len(rep["forced_gap"]) else 0.0)
            worst = max(worst, *margins)
            print(f"trial {trial} batch of {n}: {k}: batched != serial; oracle margin of the batched stream {margins[0]:.4f}, of the serial stream {margins[1]:.4f}")
print(f"{n_diff} of {n_run} batched chunks differ from their serial run; largest oracle margin among them {worst:.4f} (f16 tolerance 0.108)")
