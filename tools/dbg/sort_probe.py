import torch, time
n = 210000
x64 = torch.randint(0, 46000, (n,), device="cuda", dtype=torch.int64)
x32 = x64.to(torch.int32)
def t(fn, reps=50):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e6
print("argsort int64 stable", t(lambda: torch.argsort(x64, stable=True)))
print("argsort int32 stable", t(lambda: torch.argsort(x32, stable=True)))
print("argsort int64 unstable", t(lambda: torch.argsort(x64)))
print("sort int32 stable", t(lambda: torch.sort(x32, stable=True)))
key = (x64 << 20) | torch.arange(n, device="cuda")
print("sort composite int64 (key<<20|idx) unstable", t(lambda: torch.sort(key)))
print("to int32 + composite", t(lambda: torch.sort((x64 << 20) | torch.arange(n, device='cuda'))))
