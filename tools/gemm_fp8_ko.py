"""GPU dev tool: time the e4m3 GEMM's k loop with parts knocked out (SS_F8_KO bit mask: 1 barriers, 2 vmcnt waits, 4 DMA, 8 LDS reads, 16 MFMAs;
results are wrong by construction, only the durations mean something).  Prints ms for the FC1 / FC2 shapes, kind STORE_T.
The knock-outs are compiled out of the product library: build a variant first,
    bash tools/build_variant.sh kernels_gemm_fp8.hip -DSS_DEV_KNOCKOUTS      (-> gpurun_ab/libvariant.so, copy it over speaksense_amd/libspeaksense_hip.so on the box)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from speaksense_amd import binding, ggml_io  # noqa: E402

path = "/tmp/toy_bench.bin"
if not os.path.exists(path):
    ggml_io.write_model(path, "toy", seed=1)
eng = binding.Engine(path, dtype=binding.DTYPE_F16, max_batch=1)
out = []
for name, M, N, K in (("FC1", 12000, 5120, 1280), ("FC2", 12000, 1280, 5120), ("FC1b32", 48000, 5120, 1280)):
    _, _, ms = eng.selftest_gemm_ex(M, N, K, 0, fp8=True, reps=20)
    out.append(f"{name} {ms:.4f} ms {2.0 * M * N * K / ms / 1e9:7.1f} TF/s")
print(f"KO={os.environ.get('SS_F8_KO', '0'):>2s}  " + "   ".join(out), flush=True)
eng.close()
