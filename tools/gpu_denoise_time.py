import sys, os; sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np, time
from speaksense_amd import binding, ggml_io, synth
from oracle import denoise_oracle as d
os.makedirs("/tmp/m", exist_ok=True); ggml_io.write_model("/tmp/m/toy.bin", "toy", seed=1)
eng = binding.Engine("/tmp/m/toy.bin", max_batch=1)
for n in (80000, 480000):
    pcm = synth.speech_like(1, n)
    for _ in range(3): out, nt, nv, ms = eng.denoise_audio(pcm)
    t = time.time(); d.denoise_audio(pcm); t_cpu = time.time() - t
    frames = (n - 2048) // 512 + 1
    print(f"n={n} frames={frames} device {ms*1e3:.1f} us  ({n/16000/(ms*1e-3):.0f}x real time); numpy oracle {t_cpu*1e3:.1f} ms; algorithmic bytes {(n*4*2 + frames*2048*4*2)/1e6:.2f} MB -> {(n*8 + frames*2048*8)/ms/1e6:.1f} GB/s")
