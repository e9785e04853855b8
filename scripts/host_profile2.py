"""Host cost of the two native calls, called directly (no autograd) on the main thread, small P so the GPU is never the bottleneck."""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from manigaussian_amd import _C, _lib
from manigaussian_amd import synthetic as syn

P, F, W = int(os.environ.get("HP_P", "1000")), 32, 128
dev = torch.device("cuda:0")
sc = syn.make_scene(P, F=F, M=4, seed=0)
cam = syn.circle_cameras(8, W, W, negative_focal=True)[0]
d = {k: v.to(dev) for k, v in sc.items()}
kw = syn.camera_settings_kwargs(cam, 1, True, device=dev)
dC, dF = [t.to(dev) for t in syn.make_cotangents(W, W, F)]
e = torch.Tensor([])
L = _lib.lib()

def fwd():
    return _C.rasterize_gaussians(kw["bg"], d["means3D"], e, d["language_feature"], d["opacities"], d["scales"], d["rotations"],
                                  1.0, e, kw["viewmatrix"], kw["projmatrix"], kw["tanfovx"], kw["tanfovy"], W, W, d["shs"], 1,
                                  kw["campos"], False, False, True)

def bwd(o):
    R, color, feat, radii, geom, binning, img = o
    return _C.rasterize_gaussians_backward(kw["bg"], d["means3D"], radii, e, d["language_feature"], d["scales"], d["rotations"], 1.0, e,
                                           kw["viewmatrix"], kw["projmatrix"], kw["tanfovx"], kw["tanfovy"], dC, dF, d["shs"], 1,
                                           kw["campos"], geom, R, binning, img, False, True)

for _ in range(20):
    bwd(fwd())
torch.cuda.synchronize()
N = 500
t0 = time.perf_counter(); 
for _ in range(N): o = fwd()
torch.cuda.synchronize(); t1 = time.perf_counter()
for _ in range(N): g = bwd(o)
torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"P={P}: forward {(t1-t0)/N*1e6:.1f} us/call, backward {(t2-t1)/N*1e6:.1f} us/call")
pr = cProfile.Profile(); pr.enable()
for _ in range(N): g = bwd(fwd())
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(14)
# raw HIP API cost: empty-ish launches through the library (mark_visible = 1 launch)
m = d["means3D"]
t0 = time.perf_counter()
for _ in range(2000): _C.mark_visible(m, kw["viewmatrix"], kw["projmatrix"])
torch.cuda.synchronize(); print(f"mark_visible (1 launch + python): {(time.perf_counter()-t0)/2000*1e6:.1f} us")
x = torch.empty(1000, device=dev)
t0 = time.perf_counter()
for _ in range(2000): x.zero_()
torch.cuda.synchronize(); print(f"torch zero_ (1 launch): {(time.perf_counter()-t0)/2000*1e6:.1f} us")
t0 = time.perf_counter()
for _ in range(2000): y = torch.empty(1000, device=dev)
print(f"torch.empty: {(time.perf_counter()-t0)/2000*1e6:.1f} us")
