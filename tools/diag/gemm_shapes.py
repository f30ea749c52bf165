"""Timing of the tiled GEMM at the encoder's large-v3 shapes (ss_engine_selftest_gemm_ex, reps back-to-back launches): TF/s per shape and kind."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
from speaksense_amd import binding, ggml_io
which = sys.argv[1] if len(sys.argv) > 1 else "f16"
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 8
path = bench.model_path_for("base.en")
if not os.path.exists(path):
    ggml_io.write_model(path + ".tmp", "base.en", seed=0); os.replace(path + ".tmp", path)
eng = binding.Engine(path, dtype={"f16": binding.DTYPE_F16, "bf16": binding.DTYPE_BF16}[which], max_batch=1, n_lanes=1)
M = batch * 1500
shapes = [("FC1+GELU", M, 5120, 1280, 1), ("FC1 no GELU", M, 5120, 1280, 0), ("QK", M, 2560, 1280, 0), ("out-proj res", M, 1280, 1280, 2), ("FC2 res", M, 1280, 5120, 2), ("4096^3", 4096, 4096, 4096, 0)]
for name, m, n, k, kind in shapes:
    best = 1e9
    for _ in range(3):
        err, ref, ms = eng.selftest_gemm_ex(m, n, k, kind, False, 20)
        best = min(best, ms)
    print(f"{which} batch {batch} {name:14s} M={m} N={n} K={k}: {best*1e3:8.1f} us  {2.0*m*n*k/best/1e9:7.1f} TF/s  (err {err:.2e} of {ref:.2e})  env {os.environ.get('SS_GEMM_DBG','')} {os.environ.get('SS_GEMM_STAGGER','')}")
eng.close()
