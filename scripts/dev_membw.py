"""Achievable HBM fill / read / copy bandwidth through PyTorch kernels (development aid)."""
import torch
dev = torch.device("cuda", 0)
def t(fn, n=10):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n
for gb in (0.25, 1.0, 3.4):
    n = int(gb * 2**30 / 4)
    a = torch.empty(n, dtype=torch.int32, device=dev)
    b = torch.empty(n, dtype=torch.int32, device=dev)
    ms = t(lambda: a.fill_(1)); print(f"fill  {gb} GiB: {gb*1.0737/ms*1e3:.0f} GB/s")
    ms = t(lambda: b.copy_(a)); print(f"copy  {gb} GiB: {2*gb*1.0737/ms*1e3:.0f} GB/s (read+write)")
    ms = t(lambda: a.sum()); print(f"sum   {gb} GiB: {gb*1.0737/ms*1e3:.0f} GB/s")
    ms = t(lambda: torch.add(a, b, out=b)); print(f"add   {gb} GiB: {3*gb*1.0737/ms*1e3:.0f} GB/s (2 reads + write)")
    del a, b
