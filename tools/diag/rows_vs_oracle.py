"""Diagnostic: ss_engine_decode_rows at several row counts vs the oracle and vs the 8-row pass (large-v3, f16)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
from speaksense_amd import binding, ggml_io
from oracle import binding as orc

which = sys.argv[1] if len(sys.argv) > 1 else "f16"
dt, omode = {"f16": (binding.DTYPE_F16, orc.MODE_GGML_F16), "bf16": (binding.DTYPE_BF16, orc.MODE_BF16), "fp8": (binding.DTYPE_FP8, orc.MODE_FP8)}[which]
path = bench.model_path_for("large-v3")
if not os.path.exists(path):
    ggml_io.write_model(path + ".tmp", "large-v3", seed=0); os.replace(path + ".tmp", path)
orc.set_thread_cap(64)
om = orc.OracleModel(path)
eng = binding.Engine(path, dtype=dt, max_batch=32, n_lanes=1)
rng = np.random.default_rng(7)
n_win = 4
from speaksense_amd import synth
encs = [eng.encode(om.log_mel(synth.speech_like(200 + w)), 0) for w in range(n_win)]
for w in range(n_win):
    eng.set_encoder_window(w, encs[w])
text = rng.integers(300, 40000, (8, 8))
seqs = [[om.sot, om.sot + 1 + s, om.transcribe] + [int(t) for t in text[s][:5]] for s in range(8)]
refs = []
for s in range(8):
    ost = om.new_state(omode); ost.set_encoder(encs[s % n_win]); refs.append(ost.decode(seqs[s], 0)); ost.close()
ref32 = []
for s in range(8):
    ost = om.new_state(orc.MODE_F32); ost.set_encoder(encs[s % n_win]); ref32.append(ost.decode(seqs[s], 0)); ost.close()
def run(n_seq, slot_mul=5):
    token, pos, slot, cross, samp = [], [], [], [], []
    for s in range(n_seq):
        for i, t in enumerate(seqs[s]):
            token.append(t); pos.append(i); slot.append(slot_mul * s); cross.append(s % n_win)
        samp.append(len(token) - 1)
    return eng.decode_rows(token, pos, slot, cross, samp)
alone = [eng.decode_rows(seqs[s], list(range(8)), [5 * s] * 8, [s % n_win] * 8, [7])[0] for s in range(8)]
for n_seq in (1, 2, 4, 8):
    got = run(n_seq)
    for s in range(n_seq):
        sd = float(refs[s].std())
        print(f"{which} rows={8*n_seq} seq {s}: vs oracle {np.abs(got[s]-refs[s]).max()/sd:.2e}  alone-vs-oracle {np.abs(alone[s]-refs[s]).max()/sd:.2e}  wide-vs-alone {np.abs(got[s]-alone[s]).max()/sd:.2e}"
              f"  oracle f16-vs-f32 {np.abs(refs[s]-ref32[s]).max()/sd:.2e}  top2 gap {np.sort(refs[s])[-1]-np.sort(refs[s])[-2]:.3f} sd {sd:.2f}")
# single-token steps after the prompt (the 8-step regime of the existing stage test), sequence 1
