"""Development aid: host-to-device copy rate from pinned memory, alone."""
import time
import torch
n = 177 * 1024 * 1024
h = torch.empty(n, dtype=torch.uint8).pin_memory()
d = torch.empty(n, dtype=torch.uint8, device="cuda")
d.copy_(h, non_blocking=True); torch.cuda.synchronize()
for _ in range(2):
    t = time.perf_counter()
    for _ in range(5):
        d.copy_(h, non_blocking=True)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / 5
    print(f"{n / 1e6:.0f} MB in {dt * 1e3:.2f} ms = {n / dt / 1e9:.1f} GB/s")
