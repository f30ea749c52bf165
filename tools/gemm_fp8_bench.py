"""GPU dev tool: time the e4m3 GEMM (kernels_gemm_fp8.hip) beside the f16 GEMM at the large-v3 encoder shapes (seeded random operands, 20
back-to-back launches each, HIP events on the engine's stream).  python tools/gemm_fp8_bench.py [model.bin]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from speaksense_amd import binding, ggml_io  # noqa: E402


def main():
    path = sys.argv[1] if len(sys.argv) > 1 else "/tmp/toy_bench.bin"
    if not os.path.exists(path):
        ggml_io.write_model(path, "toy", seed=1)
    eng = binding.Engine(path, dtype=binding.DTYPE_F16, max_batch=1)
    shapes = [("QK", 12000, 2560, 1280), ("O", 12000, 1280, 1280), ("FC1", 12000, 5120, 1280), ("FC2", 12000, 1280, 5120), ("FC1 b32", 48000, 5120, 1280),
              ("FC2 b32", 48000, 1280, 5120), ("crossKV", 12000, 81920, 1280)]
    kinds16 = {"store": 0, "gelu": 1, "res_f32": 2}
    kinds8 = {"store": 0, "gelu": 1, "res_f32": 2}
    for name, M, N, K in shapes:
        for kn in ("store", "gelu", "res_f32"):
            if M * N > 12000 * 5120 * 4 and kn != "store":
                continue
            e16, r16, ms16 = eng.selftest_gemm_ex(M, N, K, kinds16[kn], fp8=False, reps=20)
            e8, r8, ms8 = eng.selftest_gemm_ex(M, N, K, kinds8[kn], fp8=True, reps=20)
            fl = 2.0 * M * N * K
            print(f"{name:8s} M={M:6d} N={N:6d} K={K:5d} {kn:8s} f16 {ms16:7.3f} ms {fl / ms16 / 1e9:7.1f} TF/s | e4m3 {ms8:7.3f} ms {fl / ms8 / 1e9:7.1f} TF/s"
                  f"  x{ms16 / ms8:4.2f}   (err/ref f16 {e16 / r16:.1e}, e4m3 {e8 / r8:.1e})", flush=True)
    eng.close()


if __name__ == "__main__":
    main()
