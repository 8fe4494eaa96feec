"""Calibration probe: achievable device-memory bandwidth vs transfer size (torch copy / fill), and the
per-kernel floor.  Numbers go to profiles/README.md to put the model kernels' GB/s in context."""
import torch, time
dev = torch.device("cuda:0")
def t(fn, n=200):
    for _ in range(10): fn()
    torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3  # us
for mb in (4, 16, 32, 64, 128, 256, 1024):
    n = mb * 1024 * 1024 // 4
    x = torch.empty(n, dtype=torch.float32, device=dev).normal_()
    y = torch.empty_like(x)
    us_c = t(lambda: y.copy_(x))
    us_f = t(lambda: y.fill_(1.0))
    us_a = t(lambda: torch.add(x, 1.0, out=y))
    print(f"{mb:5d} MB  copy {us_c:8.1f} us = {2*mb/1024/us_c*1e6/1e3:6.2f} TB/s   fill {us_f:8.1f} us = {mb/1024/us_f*1e6/1e3:6.2f} TB/s   add {us_a:8.1f} us = {2*mb/1024/us_a*1e6/1e3:6.2f} TB/s")
z = torch.empty(64, device=dev)
print("tiny kernel", t(lambda: z.fill_(0.0), 1000), "us")
