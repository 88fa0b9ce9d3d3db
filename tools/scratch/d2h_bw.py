"""D2H bandwidth of the box into pinned memory: one stream, and two streams at once (what bounds the label-copying sweep)"""
import time, torch
n = 64 << 20
d = [torch.empty(n, dtype=torch.uint8, device="cuda") for _ in range(2)]
h = [torch.empty(n, dtype=torch.uint8).pin_memory() for _ in range(2)]
s = [torch.cuda.Stream() for _ in range(2)]
for k in (1, 2):
    for rep in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for it in range(8):
            for j in range(k):
                with torch.cuda.stream(s[j]):
                    h[j].copy_(d[j], non_blocking=True)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print("%d stream(s): %.1f GB/s" % (k, 8 * k * n / dt / 1e9))
