import ctypes, os, sys, torch
sys.path.insert(0, os.getcwd())
from drt_amd import _lib
n = 72 * 1024 * 1024
dev = "cuda"
torch.manual_seed(0)
oo = torch.randn(n, 3, dtype=torch.float64, device=dev); od = torch.randn(n, 3, dtype=torch.float64, device=dev)
sp = torch.randn(n, 3, dtype=torch.float64, device=dev)
mask = (torch.rand(n, device=dev) < 0.025).to(torch.uint8).unsqueeze(1).expand(n, 3).contiguous()
valid = (torch.rand(n, device=dev) < 0.9).to(torch.uint8)
loss = torch.zeros((), dtype=torch.float64, device=dev)
g = torch.empty_like(od); rows = torch.empty(n, dtype=torch.int32, device=dev); nrows = torch.zeros(1, dtype=torch.int32, device=dev)
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
def run(gp, lp):
    nrows.zero_()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(5):
        nrows.zero_(); a.record()
        _lib.check(_lib.lib().drt_ray_loss(oo.data_ptr(), od.data_ptr(), mask.data_ptr(), sp.data_ptr(), valid.data_ptr(), n, loss.data_ptr(),
                                           g.data_ptr() if gp else None, rows.data_ptr() if lp else None, nrows.data_ptr() if lp else None, st))
        b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    return min(ts)
print("dense grad + list : %.3f ms" % run(True, True))
print("dense grad only   : %.3f ms" % run(True, False))
print("list only         : %.3f ms" % run(False, True))
print("neither           : %.3f ms" % run(False, False))
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record(); g.zero_(); b.record(); torch.cuda.synchronize(); print("torch zero_ of the dense grad: %.3f ms" % a.elapsed_time(b))
