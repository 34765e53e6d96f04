import torch, time
x = torch.empty(2_200_000_000 // 8, dtype=torch.float64, device="cuda")
y = torch.empty_like(x)
def t(f, n=10):
    f(); torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter()-t0)/n
dt = t(lambda: x.zero_()); print("zero_ 2.2GB", dt*1e3, "ms", 2.2/dt/1e3, "TB/s")
dt = t(lambda: y.copy_(x)); print("copy 2.2GB->2.2GB", dt*1e3, "ms", 4.4/dt/1e3, "TB/s")
dt = t(lambda: x.sum()); print("read 2.2GB", dt*1e3, "ms", 2.2/dt/1e3, "TB/s")
