import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from manigaussian_amd import GaussianRasterizationSettings, GaussianRasterizer, check_status
from manigaussian_amd import synthetic as syn
P, F, W = 100000, 32, 128
dev = torch.device("cuda:0")
sc = syn.make_scene(P, F=F, M=4, seed=0)
cam = syn.circle_cameras(8, W, W, negative_focal=True)[0]
params = {k: v.to(dev).requires_grad_(True) for k, v in sc.items()}
plist = list(params.values())
rast = GaussianRasterizer(GaussianRasterizationSettings(**syn.camera_settings_kwargs(cam, 1, True, device=dev)))
dC, dF = [t.to(dev) for t in syn.make_cotangents(W, W, F)]
m2 = torch.zeros(P, 3, device=dev, requires_grad=True)
def step():
    c, f, r = rast(means3D=params["means3D"], means2D=m2, opacities=params["opacities"], shs=params["shs"],
                   language_feature_precomp=params["language_feature"], scales=params["scales"], rotations=params["rotations"])
    return torch.autograd.grad([c, f], plist, [dC, dF])
for _ in range(5):
    step(); check_status(dev)
side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    step()
torch.cuda.current_stream().wait_stream(side)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    out = step()
for _ in range(30):
    g.replay()
torch.cuda.synchronize()
for trial in range(3):
    K = 20
    torch.cuda.synchronize()
    ts = [time.perf_counter()]
    evs = []
    for i in range(K):
        g.replay()
        ts.append(time.perf_counter())
    t_enq = time.perf_counter()
    ev = torch.cuda.Event(); ev.record()
    while not ev.query():
        pass
    t_done = time.perf_counter()
    torch.cuda.synchronize()
    t_sync = time.perf_counter()
    d = [(ts[i + 1] - ts[i]) * 1e6 for i in range(K)]
    print(f"trial {trial}: per-replay host us: first {d[0]:.1f}, second {d[1]:.1f}, mean rest {sum(d[2:]) / (K - 2):.1f}; enqueue done at {(t_enq - ts[0]) * 1e6:.0f} us, GPU done at {(t_done - ts[0]) * 1e6:.0f} us "
          f"({(t_done - ts[0]) * 1e3 / K:.4f} ms/step), sync returned at {(t_sync - ts[0]) * 1e6:.0f} us")
# eager single-thread for comparison
torch.autograd.set_multithreading_enabled(False)
for _ in range(30):
    step()
for trial in range(2):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(20):
        step()
    ev = torch.cuda.Event(); ev.record()
    while not ev.query():
        pass
    t1 = time.perf_counter()
    print(f"eager-st 20 steps: {(t1 - t0) * 1e3 / 20:.4f} ms/step")
