"""Host-to-device bandwidth with 1..4 concurrent copy streams (pinned source): is the 29 GB/s the
advice upload sees the link or one SDMA engine?"""
import time
import torch

n = 1 << 30
src = [torch.empty(n // 4, dtype=torch.uint8).pin_memory() for _ in range(4)]
dst = [torch.empty(n // 4, dtype=torch.uint8, device="cuda") for _ in range(4)]
for ns in (1, 2, 4):
    streams = [torch.cuda.Stream() for _ in range(ns)]
    for rep in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(4):
            with torch.cuda.stream(streams[i % ns]):
                dst[i].copy_(src[i], non_blocking=True)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    print(f"{ns} stream(s): {n / dt / 1e9:.1f} GB/s")
