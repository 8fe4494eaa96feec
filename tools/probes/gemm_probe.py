import torch, time
torch.manual_seed(0)
for M in (38700, 62000):
    for dt in (torch.float16, torch.bfloat16):
        A = torch.randn(M, 3968, device="cuda", dtype=dt); W = torch.randn(3968, 256, device="cuda", dtype=dt)
        for _ in range(5): C = A @ W
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): C = A @ W
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 20
        print(f"hipBLASLt/rocBLAS via torch: M={M} K=3968 N=256 {dt}: {us:.1f} us  {2*M*3968*256/us/1e6:.0f} TFLOP/s")
