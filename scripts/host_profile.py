"""Host-side (Python + HIP runtime) cost of one fwd+bwd step at the headline config: cProfile over N steps."""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from manigaussian_amd import GaussianRasterizationSettings, GaussianRasterizer
from manigaussian_amd import synthetic as syn

P, F, W = int(os.environ.get("HP_P", "100000")), 32, 128
dev = torch.device("cuda:0")
sc = syn.make_scene(P, F=F, M=4, seed=0)
cam = syn.circle_cameras(8, W, W, negative_focal=True)[0]
d = {k: v.to(dev).requires_grad_(True) for k, v in sc.items()}
plist = list(d.values())
m2d = torch.zeros(P, 3, device=dev, requires_grad=True)
dC, dF = [t.to(dev) for t in syn.make_cotangents(W, W, F)]
rast = GaussianRasterizer(GaussianRasterizationSettings(**syn.camera_settings_kwargs(cam, 1, True, device=dev)))


def step():
    c, f, r = rast(d["means3D"], m2d, d["opacities"], shs=d["shs"], language_feature_precomp=d["language_feature"],
                   scales=d["scales"], rotations=d["rotations"])
    return torch.autograd.grad([c, f], plist, [dC, dF])


for _ in range(20):
    step()
torch.cuda.synchronize()
N = 300
t0 = time.perf_counter()
for _ in range(N):
    step()
t_enq = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print(f"enqueue {t_enq / N * 1e6:.1f} us/step, with final sync {t_all / N * 1e6:.1f} us/step")
pr = cProfile.Profile()
pr.enable()
for _ in range(N):
    step()
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
