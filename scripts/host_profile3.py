"""Host cost of one training step through the PUBLIC autograd API (small P: the GPU is never the bottleneck):
us per step with torch's default autograd threading and with set_multithreading_enabled(False), then a cProfile of the latter."""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from manigaussian_amd import GaussianRasterizationSettings, GaussianRasterizer, check_status
from manigaussian_amd import synthetic as syn

P, F, W = int(os.environ.get("HP_P", "1000")), 32, 128
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


for _ in range(50):
    step()
check_status(dev)
N = 2000
for name, mt in (("default autograd threading", True), ("set_multithreading_enabled(False)", False)):
    torch.autograd.set_multithreading_enabled(mt)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(N):
        step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    print(f"P={P} {name}: {(t1 - t0) / N * 1e6:.1f} us of host time per step (enqueue only), {(time.perf_counter() - t0) / N * 1e6:.1f} us incl. drain")
torch.autograd.set_multithreading_enabled(False)
pr = cProfile.Profile()
pr.enable()
for _ in range(N):
    step()
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
