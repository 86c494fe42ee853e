"""Read-bandwidth yardstick on the box: torch.sum over fp64 tensors larger than the Infinity Cache (a library reduction that
only reads), next to the NN sweep's rate.  Diagnostic."""
import torch, time
for mb in (403, 1612):
    n = mb * 1024 * 1024 // 8
    x = torch.rand(n, dtype=torch.float64, device="cuda")
    for _ in range(3):
        x.sum()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20):
        x.sum()
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 20
    print(f"torch.sum fp64 {mb} MB: {ms*1e3:.1f} us = {n*8/ms/1e6:.0f} GB/s", flush=True)
    y = torch.empty_like(x)
    a.record()
    for _ in range(20):
        y.copy_(x)
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 20
    print(f"copy {mb} MB: {ms*1e3:.1f} us = read+write {2*n*8/ms/1e6:.0f} GB/s", flush=True)
