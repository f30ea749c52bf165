"""Does freeing an engine from the garbage collector (Engine.__del__) some time after heavy use crash?  (r04_h: a failed soak test leaked its engine;
the process segfaulted seconds later inside another test's subprocess.communicate -- where the GC happened to run.)"""
import gc
import os
import subprocess
import sys
import tempfile
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import faulthandler

faulthandler.enable()
from speaksense_amd import binding, ggml_io, synth

path = os.path.join(tempfile.mkdtemp(), "toy.bin")
ggml_io.write_model(path, "toy", seed=1)


def load(eng, seconds):
    t_end = time.time() + seconds
    lens = [0.5, 3.0, 12.0, 30.1, 65.0]

    def worker(w):
        k = 0
        while time.time() < t_end:
            ses = [eng.new_session() for _ in range(3)]
            ts = [s.submit(synth.speech_like(w + k, int(16000 * lens[(w + k + i) % 5])),
                           binding.default_params(language="en", temperature_inc=0.2 if (k + i) % 3 == 0 else 0.0)) for i, s in enumerate(ses)]
            for i, (s, t) in enumerate(zip(ses, ts)):
                if i == 1:
                    s.close(); eng.L.ss_wait(t)
                else:
                    s.wait(t); s.close()
            k += 1
    th = [threading.Thread(target=worker, args=(w,)) for w in range(8)]
    for t in th:
        t.start()
    for t in th:
        t.join()


for mode in ("explicit close", "gc after a child process", "gc at once"):
    eng = binding.Engine(path, max_batch=8, n_lanes=3)
    load(eng, float(os.environ.get("REPRO_SECONDS", "20")))
    print(mode, "graphs / mem", eng.mem_info(), flush=True)
    if mode == "explicit close":
        eng.close()
    elif mode == "gc at once":
        del eng; gc.collect()
    else:
        ref = [eng]; del eng
        subprocess.run([sys.executable, "-c", "import time; time.sleep(2)"], capture_output=True)
        ref.clear(); gc.collect()
        subprocess.run([sys.executable, "-c", "import time; time.sleep(1)"], capture_output=True)
    print(mode, "ok", flush=True)
print("no crash")
