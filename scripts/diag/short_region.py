"""Where do the ~0.25 ms go that a 20-step EAGER region costs over 20 x the long-run step time?  (bench.py's driver form)
Prints, for consecutive 20-step regions through the package defaults: wall time, the GPU span between a hipEvent recorded before
the first step and one after the last, and the host time stamps of every step's enqueue."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from manigaussian_amd import GaussianRasterizationSettings, GaussianRasterizer, check_status
from manigaussian_amd import synthetic as syn

dev = torch.device("cuda:0")
P, F, W = 100000, 32, 128
sc = syn.make_scene(P, F=F, M=4, seed=0)
params = {k: v.to(dev).requires_grad_(True) for k, v in sc.items()}
means2D = torch.zeros(P, 3, device=dev, requires_grad=True)
cam = syn.circle_cameras(8, W, W, negative_focal=True)[0]
rast = GaussianRasterizer(GaussianRasterizationSettings(**syn.camera_settings_kwargs(cam, 1, True, device=dev)))
dC, dF = (t.to(dev) for t in syn.make_cotangents(W, W, F, seed=1))
plist = list(params.values())


def step():
    c, f, r = rast(means3D=params["means3D"], means2D=means2D, opacities=params["opacities"], shs=params["shs"],
                   language_feature_precomp=params["language_feature"], scales=params["scales"], rotations=params["rotations"])
    return torch.autograd.grad([c, f], plist, [dC, dF])


for _ in range(300):
    step()
torch.cuda.synchronize()
K = int(sys.argv[1]) if len(sys.argv) > 1 else 20
for rep in range(6):
    torch.cuda.synchronize()
    if rep >= 3:
        time.sleep(0.002 * (rep - 2))  # an idle gap before the region, like the bench's brackets
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    stamps = []
    t0 = time.perf_counter()
    e0.record()
    for i in range(K):
        step()
        stamps.append(time.perf_counter() - t0)
    e1.record()
    t_enq = time.perf_counter() - t0
    while not e1.query():
        pass
    t_done = time.perf_counter() - t0
    torch.cuda.synchronize()
    print(f"region {rep}: wall {t_done * 1e3:.3f} ms = {t_done / K * 1e3:.4f} per step; host enqueue done at {t_enq * 1e3:.3f} ms; GPU span "
          f"{e0.elapsed_time(e1):.3f} ms; first steps enqueued at {[round(s * 1e3, 3) for s in stamps[:6]]} ms")
check_status(dev)
