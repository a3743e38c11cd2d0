import torch, time
n = 12 << 30
x = torch.empty(n, dtype=torch.uint8, device="cuda")
for i in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter(); x.fill_(i); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("fill", n / dt / 1e9, "GB/s")
y = torch.empty(n, dtype=torch.uint8, device="cuda")
for i in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter(); y.copy_(x); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("copy", n / dt / 1e9, "GB/s (each way)")
del y
for i in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter(); z = torch.empty(n, dtype=torch.uint8, device="cuda"); z.fill_(1); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("alloc+fill", n / dt / 1e9, "GB/s"); del z; torch.cuda.empty_cache()
