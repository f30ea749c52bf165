"""Dev tool (not product): what the vendor library reaches on the encoder GEMM shapes, as a yardstick for gemm256_kernel."""
import torch, time
torch.backends.cuda.matmul.allow_fp16_reduced_precision_reduction = False
dev = "cuda"
for (M, N, K, name) in [(12000, 5120, 1280, "FC1"), (12000, 1280, 5120, "FC2"), (12000, 3840, 1280, "QKV"), (12000, 1280, 1280, "O"), (12000, 81920, 1280, "crossKV")]:
    a = torch.randn(M, K, device=dev, dtype=torch.float16)
    w = torch.randn(N, K, device=dev, dtype=torch.float16)
    for _ in range(5): c = a @ w.t()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 30
    e0.record()
    for _ in range(reps): c = a @ w.t()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print(f"{name:8s} M={M} N={N} K={K}: {ms*1e3:8.1f} us  {2.0*M*N*K/ms/1e9:7.1f} TFLOP/s", flush=True)
