"""How fast does the vendor library (torch.mm -> hipBLASLt / rocBLAS) run the DiT's GEMM shapes?  A yardstick for csrc/gemm_bf16.hip only:
the product path never calls it."""
import json, torch, torch.nn.functional as F
shapes = [(8192, 8960, 1536), (8192, 1536, 8960), (8192, 3072, 1536), (8192, 1536, 1536), (1536, 8192, 1536), (8192, 8192, 8192)]
for M, N, K in shapes:
    a = torch.randn(M, K, device="cuda").bfloat16(); w = (torch.randn(N, K, device="cuda") / K ** 0.5).bfloat16()
    b = torch.randn(N, device="cuda").bfloat16()
    for name, fn in (("mm", lambda: a @ w.t()), ("linear+bias", lambda: F.linear(a, w, b))):
        for _ in range(5): fn()
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20): fn()
            e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / 20)
        print(json.dumps(dict(M=M, N=N, K=K, op=name, us=round(best * 1e3, 1), tflops=round(2 * M * N * K / best / 1e9))), flush=True)
