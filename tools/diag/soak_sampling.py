"""dev diagnostic (GPU box): which parameter family produces the 1-in-100 000 soak mismatch of r05_ak (a fresh session's chunk differing from the serial run)?
8 threads submit single chunks on fresh sessions of one 3-lane engine; per variant: chunks checked / mismatches, and for a mismatch where the traces part."""
import os, sys, threading, time, random, tempfile
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from speaksense_amd import binding, ggml_io, synth

seconds = float(os.environ.get("SECONDS_", "240"))
model = os.environ.get("MODEL_", "toy")          # MODEL_=wide2 DTYPE_=fp8: the e4m3 engine (needs d % 256 == 0)
dtype = {"f16": binding.DTYPE_F16, "bf16": binding.DTYPE_BF16, "fp8": binding.DTYPE_FP8}[os.environ.get("DTYPE_", "f16")]
path = os.path.join(tempfile.mkdtemp(), model + ".bin")
ggml_io.write_model(path, model, seed=1)
eng = binding.Engine(path, dtype=dtype, max_batch=8, n_lanes=3)
print("model", model, "dtype", os.environ.get("DTYPE_", "f16"), flush=True)
variants = {
    "ctx256_ladder": dict(audio_ctx=256, temperature_inc=0.2),
    "ladder_forced": dict(temperature_inc=0.2, logprob_thold=0.0),
    "ctx256_greedy": dict(audio_ctx=256, temperature_inc=0.0),
    "ctx752_ladder_forced": dict(audio_ctx=752, temperature_inc=0.2, logprob_thold=0.0),
    "greedy": dict(temperature_inc=0.0),
}
lengths = [3.0, 12.0, 29.9, 30.1, 44.0]
audio = {(sd, ln): synth.speech_like(sd, int(16000 * ln)) for sd in (1, 2, 3) for ln in lengths}
P = lambda v: binding.default_params(**dict(dict(language="en"), **variants[v]))
key = lambda r: (list(int(t) for t in r["trace"]), [(s["t0"], s["t1"], s["text"]) for s in r["segments"]], r["n_fail"])
serial, lock = {}, threading.Lock()
def expected(k):
    with lock:
        if k not in serial:
            serial[k] = key(eng.new_session().transcribe(audio[k[0]], P(k[1])))
        return serial[k]
for k in [((sd, ln), v) for sd in (1, 2, 3) for ln in lengths for v in variants]:
    expected(k)
print("serial results ready; fallbacks per variant:", {v: sum(serial[((sd, ln), v)][2] for sd in (1, 2, 3) for ln in lengths) for v in variants}, flush=True)
stats = {v: [0, 0] for v in variants}
found = []
t_end = time.time() + seconds
def worker(w):
    rnd = random.Random(7000 + w)
    while time.time() < t_end:
        n = rnd.randint(1, 6)
        items = []
        for _ in range(n):
            k = ((rnd.choice((1, 2, 3)), rnd.choice(lengths)), rnd.choice(list(variants)))
            s = eng.new_session()
            items.append((s, s.submit(audio[k[0]], P(k[1])), k))
        rnd.shuffle(items)
        for s, t, k in items:
            got = key(s.wait(t))
            exp = expected(k)
            with lock:
                stats[k[1]][0] += 1
                if got != exp:
                    stats[k[1]][1] += 1
                    a, b = got[0], exp[0]
                    i = next((i for i in range(min(len(a), len(b))) if a[i] != b[i]), min(len(a), len(b)))
                    found.append((k, i, len(a), len(b), a[max(0, i - 2): i + 3], b[max(0, i - 2): i + 3], got[2], exp[2]))
            s.close()
th = [threading.Thread(target=worker, args=(w,)) for w in range(8)]
[t.start() for t in th]; [t.join() for t in th]
print("checked / mismatches per variant:", stats)
for f in found[:20]:
    print("MISMATCH", f)
eng.close()
