#!/usr/bin/env python3
"""How does this runtime move pinned host memory to the device: DMA engine or copy kernel?  Five 1 GiB copies and five
64 MiB copies (linear), and a strided 2-D one, for use under rocprofv3 --kernel-trace --memory-copy-trace --stats."""
import time
import torch
dev = torch.device("cuda", 0)
src = torch.empty(1 << 30, dtype=torch.uint8, pin_memory=True)
dst = torch.empty(1 << 30, dtype=torch.uint8, device=dev)
torch.cuda.synchronize()
for size in (1 << 30, 1 << 26, 1 << 22):
    dst[:size].copy_(src[:size], non_blocking=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        dst[:size].copy_(src[:size], non_blocking=True)
    torch.cuda.synchronize()
    print("%d MiB linear: %.1f GB/s" % (size >> 20, 5 * size / (time.perf_counter() - t0) / 1e9))
