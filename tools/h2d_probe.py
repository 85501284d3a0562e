#!/usr/bin/env python3
"""Host-to-device copy rate of one batch of frames (32 x 720p BGR = 88 MB, pinned) split over 1 / 2 / 4 HIP streams."""
import time
import torch

n = 32 * 720 * 1280 * 3
host = torch.empty(n, dtype=torch.uint8).pin_memory()
host.random_(0, 255)
dev = torch.empty(n, dtype=torch.uint8, device="cuda")
for parts in (1, 2, 4, 8):
    streams = [torch.cuda.Stream() for _ in range(parts)]
    step = n // parts
    for rep in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for it in range(10):
            for k, s in enumerate(streams):
                with torch.cuda.stream(s):
                    dev[k * step:(k + 1) * step].copy_(host[k * step:(k + 1) * step], non_blocking=True)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 10
    assert torch.equal(dev.cpu(), host)
    print(f"{parts} stream(s): {dt * 1e3:.2f} ms per batch = {n / dt / 1e9:.1f} GB/s")
