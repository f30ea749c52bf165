"""How does a `write_model(**NATURAL)` model (speaksense_amd/ggml_io.py natural_tensor) behave under whisper.cpp's FULL decoding rules?
Transcribes N synthetic chunks with the reference's parameters (whisper.rs:131-173: ladder on, best_of 5) and prints per chunk: tokens, windows,
fallbacks; then the summary VERDICT r03 #5 asks for (share of windows that stay at temperature 0, spread of the lengths, distinct streams).
  GPU box:    python tools/natural_preset_stats.py --model large-v3 --n 32 [knob=value ...]        (the HIP engine)
  build box:  python tools/natural_preset_stats.py --model wide2 --n 8 --oracle [knob=value ...]   (the CPU oracle; small shapes only)"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from speaksense_amd import ggml_io, synth   # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="large-v3")
    ap.add_argument("--n", type=int, default=32)
    ap.add_argument("--seed0", type=int, default=100)
    ap.add_argument("--oracle", action="store_true")
    ap.add_argument("--model-seed", type=int, default=0)
    ap.add_argument("--dtype", default="f16")
    ap.add_argument("knobs", nargs="*")
    a = ap.parse_args()
    kw = dict(ggml_io.NATURAL)
    for k in a.knobs:
        name, v = k.split("=")
        kw[name] = type(kw[name])(float(v))
    d = os.environ.get("SS_MODEL_DIR", "/tmp/ss_models")
    os.makedirs(d, exist_ok=True)
    tag = "-".join(f"{k}{v}" for k, v in sorted(kw.items()) if ggml_io.NATURAL.get(k) != v)
    path = os.path.join(d, f"ggml-{a.model}-natural{('-' + tag) if tag else ''}-s{a.model_seed}.bin")
    if not os.path.exists(path):
        t = time.time()
        ggml_io.write_model(path + ".tmp", a.model, seed=a.model_seed, **kw)
        os.replace(path + ".tmp", path)
        print(f"wrote {path} in {time.time() - t:.0f} s", file=sys.stderr)
    pcms = [synth.speech_like(a.seed0 + i) for i in range(a.n)]
    t0 = time.time()
    if a.oracle:
        from oracle import binding as orc
        orc.set_thread_cap(64)
        om = orc.OracleModel(path)
        res = [om.new_state(orc.MODE_GGML_F16).full(p, orc.default_params(language="en")) for p in pcms]
        n_win = [r["n_encode"] for r in res]
    else:
        from speaksense_amd import binding
        eng = binding.Engine(path, dtype={"f16": binding.DTYPE_F16, "bf16": binding.DTYPE_BF16, "fp8": binding.DTYPE_FP8}[a.dtype], max_batch=32, n_lanes=2)
        ses = [eng.new_session() for _ in pcms]
        tickets = [s.submit(p, binding.default_params(language="en")) for s, p in zip(ses, pcms)]
        res = [s.wait(t) for s, t in zip(ses, tickets)]
        n_win = [r["n_windows"] for r in res]
    dt = time.time() - t0
    lens = [len(r["tokens"]) for r in res]
    fails = [r["n_fail"] for r in res]
    streams = {tuple(int(t) for t in r["tokens"]) for r in res}
    for i, r in enumerate(res):
        print(f"chunk {i:3d}: {lens[i]:4d} tokens, {n_win[i]} windows, {fails[i]} fallbacks, {len(r['segments'])} segments")
    clean = sum(1 for f in fails if f == 0)
    print(f"{a.model} natural {tag or '(defaults)'}: {a.n} chunks in {dt:.1f} s; windows {sum(n_win)}, fallbacks {sum(fails)} "
          f"(chunks without any: {clean}/{a.n}; windows at t = 0: >= {100.0 * (sum(n_win) - sum(fails)) / max(1, sum(n_win)):.0f} %); tokens per chunk min / median / max "
          f"{min(lens)} / {int(np.median(lens))} / {max(lens)}; distinct streams {len(streams)}/{a.n}")


if __name__ == "__main__":
    main()
