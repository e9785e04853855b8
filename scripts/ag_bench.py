import time, torch
dev = torch.device("cuda:0")
class Dummy(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b, c, d, e, f, g, h, i, s):
        o1 = torch.empty(3, 128, 128, device=dev); o2 = torch.empty(32, 128, 128, device=dev); o3 = torch.empty(1000, dtype=torch.int32, device=dev)
        ctx.save_for_backward(a, b, c, d, e, f, g, h, i, o3)
        return o1, o2, o3
    @staticmethod
    def backward(ctx, g1, g2, g3):
        a, b, c, d, e, f, g, h, i, o3 = ctx.saved_tensors
        return (torch.empty_like(a), torch.empty_like(b), torch.empty_like(c), None, torch.empty_like(e), torch.empty_like(f), torch.empty_like(g), torch.empty_like(h), None, None)
P = 1000
mk = lambda *s: torch.randn(*s, device=dev, requires_grad=True)
a, b, c, e, f, g, h = mk(P, 3), mk(P, 3), mk(P, 4, 3), mk(P, 32), mk(P, 1), mk(P, 3), mk(P, 4)
d = torch.Tensor([]); i = torch.Tensor([])
dC = torch.randn(3, 128, 128, device=dev); dF = torch.randn(32, 128, 128, device=dev)
plist = [a, c, e, f, g, h]
def step():
    o1, o2, o3 = Dummy.apply(a, b, c, d, e, f, g, h, i, None)
    return torch.autograd.grad([o1, o2], plist, [dC, dF])
for _ in range(50): step()
torch.cuda.synchronize()
N = 2000
t0 = time.perf_counter()
for _ in range(N): step()
torch.cuda.synchronize()
print(f"dummy autograd step: {(time.perf_counter()-t0)/N*1e6:.1f} us")
t0 = time.perf_counter()
for _ in range(N):
    with torch.no_grad():
        o = Dummy.apply(a, b, c, d, e, f, g, h, i, None)
print(f"apply under no_grad: {(time.perf_counter()-t0)/N*1e6:.1f} us")
t0 = time.perf_counter()
for _ in range(N): o = Dummy.apply(a, b, c, d, e, f, g, h, i, None)
print(f"apply with grad: {(time.perf_counter()-t0)/N*1e6:.1f} us")
with torch.autograd.set_multithreading_enabled(False):
    for _ in range(50): step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(N): step()
    torch.cuda.synchronize()
    print(f"dummy autograd step, multithreading off: {(time.perf_counter()-t0)/N*1e6:.1f} us")
import os
print("cpus", os.cpu_count(), "threads", torch.get_num_threads())
