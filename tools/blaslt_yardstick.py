"""Dev tool (not product): what the vendor library (hipBLASLt through torch.matmul) reaches on the encoder GEMM shapes, as a yardstick for
gemm256k64_kernel.  Operands drawn like tools/gemm_bench.cpp draws them (A uniform [-1, 1), W uniform [-0.05, 0.05)): the chip's clock under an
MFMA loop depends on the bits that toggle, so a yardstick on other data measures another power point."""
import os
import torch
torch.backends.cuda.matmul.allow_fp16_reduced_precision_reduction = False
dev = "cuda"
for (M, N, K, name) in [(12000, 5120, 1280, "FC1"), (12000, 1280, 5120, "FC2"), (12000, 3840, 1280, "QKV"), (12000, 2560, 1280, "QK"), (12000, 1280, 1280, "O"),
                        (12000, 81920, 1280, "crossKV"), (48000, 5120, 1280, "FC1x4"), (48000, 1280, 5120, "FC2x4"), (48000, 1280, 1280, "Ox4"), (48000, 2560, 1280, "QKx4")]:
    a = (torch.rand(M, K, device=dev) * 2 - 1).to(torch.float16)
    w = ((torch.rand(N, K, device=dev) * 2 - 1) * 0.05).to(torch.float16)
    for _ in range(5):
        c = a @ w.t()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = int(os.environ.get("SS_YARD_REPS", "20"))
    e0.record()
    for _ in range(reps):
        c = a @ w.t()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print(f"hipBLASLt {name:8s} M={M} N={N} K={K}: {ms:8.3f} ms  {2.0*M*N*K/ms/1e9:7.1f} TF/s", flush=True)
    del a, w, c
