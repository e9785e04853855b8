"""Run fwd+bwd steps of the headline config with library options given as key=value args (for rocprofv3)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from manigaussian_amd import _lib, GaussianRasterizationSettings, GaussianRasterizer
from manigaussian_amd import synthetic as syn

opts = dict(P=100000, F=32, W=128, steps=20, cam=0)
for a in sys.argv[1:]:
    k, v = a.split("=")
    if k in opts:
        opts[k] = int(v)
    else:
        _lib.set_option(k, int(v))
P, F, W = opts["P"], opts["F"], opts["W"]
dev = torch.device("cuda:0")
sc = syn.make_scene(P, F=F, M=4, seed=0)
cam = syn.circle_cameras(8, W, W, negative_focal=True)[opts["cam"]]
d = {k: v.to(dev).requires_grad_(True) for k, v in sc.items()}
m2d = torch.zeros(P, 3, device=dev, requires_grad=True)
dC, dF = [t.to(dev) for t in syn.make_cotangents(W, W, F)]
rast = GaussianRasterizer(GaussianRasterizationSettings(**syn.camera_settings_kwargs(cam, 1, True, device=dev)))
def step():
    c, f, r = rast(d["means3D"], m2d, d["opacities"], shs=d["shs"], language_feature_precomp=d["language_feature"],
                   scales=d["scales"], rotations=d["rotations"])
    torch.autograd.backward([c, f], [dC, dF])
    for t in d.values():
        t.grad = None
    m2d.grad = None
for _ in range(3):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(opts["steps"]):
    step()
torch.cuda.synchronize()
print(f"{sys.argv[1:]}: {(time.perf_counter() - t0) / opts['steps'] * 1e3:.3f} ms/step")
