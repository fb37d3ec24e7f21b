#!/usr/bin/env python3
"""What a pure WRITE stream achieves on this GPU (context for the compat kernel's HBM fraction): torch fill_ / copy_ of
100 MB and 1.6 GB buffers, timed with events."""
import torch, time
dev = torch.device("cuda", 0)
def timeit(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3
for mb in (100, 400, 1600):
    x = torch.empty(mb * 1000 * 1000 // 4, dtype=torch.float32, device=dev)
    y = torch.empty_like(x)
    t = timeit(lambda: x.fill_(1.0)); print(f"fill  {mb:5d} MB: {t*1e6:8.1f} us  {mb*1e6/t/1e12:.2f} TB/s written")
    t = timeit(lambda: x.zero_());    print(f"zero  {mb:5d} MB: {t*1e6:8.1f} us  {mb*1e6/t/1e12:.2f} TB/s written")
    t = timeit(lambda: y.copy_(x));   print(f"copy  {mb:5d} MB: {t*1e6:8.1f} us  {2*mb*1e6/t/1e12:.2f} TB/s read+written")
    t = timeit(lambda: x.sum());      print(f"read  {mb:5d} MB: {t*1e6:8.1f} us  {mb*1e6/t/1e12:.2f} TB/s read")
