"""What a plain device copy reaches on this box (the practical HBM ceiling the streaming kernels are held against)."""
import time
import torch
for mb in (256, 1024, 2048):
    n = mb * 1024 * 1024 // 4
    x = torch.randn(n, device='cuda'); y = torch.empty_like(x)
    for _ in range(3): y.copy_(x)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): y.copy_(x)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
    print(f'copy {mb} MB: {2 * n * 4 / dt / 1e12:.2f} TB/s (read+write)', flush=True)
    for _ in range(3): y.fill_(1.0)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): y.fill_(1.0)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
    print(f'fill {mb} MB: {n * 4 / dt / 1e12:.2f} TB/s (write)', flush=True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): s = x.sum()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
    print(f'sum  {mb} MB: {n * 4 / dt / 1e12:.2f} TB/s (read)', flush=True)
