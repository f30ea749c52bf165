"""8-phase GEMM knock-outs (results WRONG, timing only): prints us / TF for 4096^3 and FC1-noGELU under the current SS_GEMM_KO."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
from speaksense_amd import binding, ggml_io
path = bench.model_path_for("base.en")
if not os.path.exists(path):
    ggml_io.write_model(path + ".tmp", "base.en", seed=0); os.replace(path + ".tmp", path)
eng = binding.Engine(path, dtype=binding.DTYPE_F16, max_batch=1, n_lanes=1)
out = []
for name, m, n, k, kind in [("4096^3", 4096, 4096, 4096, 0), ("FC1 noGELU", 48000, 5120, 1280, 0)]:
    best = min(eng.selftest_gemm_ex(m, n, k, kind, False, 20)[2] for _ in range(3))
    out.append(f"{name}: {best*1e3:7.1f} us {2.0*m*n*k/best/1e9:7.1f} TF")
print(f"P8={os.environ.get('SS_GEMM_P8','-')} KO={os.environ.get('SS_GEMM_KO','0'):>2s}  " + "   ".join(out), flush=True)
eng.close()
