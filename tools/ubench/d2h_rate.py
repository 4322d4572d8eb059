import torch, time
dev = torch.device("cuda", 0)
for mb in (16, 47, 100, 256):
    n = mb * 1000 * 1000
    src = torch.empty(n, dtype=torch.uint8, device=dev).random_()
    dst = torch.empty(n, dtype=torch.uint8, pin_memory=True)
    for _ in range(3):
        dst.copy_(src, non_blocking=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        dst.copy_(src, non_blocking=True)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 10
    print("D2H %4d MB: %.3f ms  %.1f GB/s" % (mb, dt * 1e3, n / dt / 1e9))
    src2 = torch.empty(n, dtype=torch.uint8, pin_memory=True)
    d2 = torch.empty(n, dtype=torch.uint8, device=dev)
    for _ in range(3):
        d2.copy_(src2, non_blocking=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        d2.copy_(src2, non_blocking=True)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 10
    print("H2D %4d MB: %.3f ms  %.1f GB/s" % (mb, dt * 1e3, n / dt / 1e9))
