"""Multi-view batch vs per-view calls at the headline shape: ms per view, fwd+bwd."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from manigaussian_amd import GaussianRasterizationSettings, GaussianRasterizer, GaussianRasterizerBatch, _lib
from manigaussian_amd import synthetic as syn

P, F, W = int(os.environ.get("BV_P", "100000")), 32, int(os.environ.get("BV_W", "128"))
dev = torch.device("cuda:0")
torch.autograd.set_multithreading_enabled(False)
sc = syn.make_scene(P, F=F, M=4, seed=0)
d = {k: v.to(dev).requires_grad_(True) for k, v in sc.items()}
plist = list(d.values())
for V in [int(x) for x in os.environ.get("BV_V", "1,2,4,8,16").split(",")]:
    cams = syn.circle_cameras(max(V, 8), W, W, negative_focal=True)[:V]
    sets = [GaussianRasterizationSettings(**syn.camera_settings_kwargs(c, 1, True, device=dev)) for c in cams]
    g = torch.Generator().manual_seed(1)
    dC, dF = torch.randn(V, 3, W, W, generator=g).to(dev), torch.randn(V, F, W, W, generator=g).to(dev)
    batch = GaussianRasterizerBatch(sets)
    singles = [GaussianRasterizer(s) for s in sets]

    def step_batch():
        c, f, r = batch(d["means3D"], None, d["opacities"], shs=d["shs"], language_feature_precomp=d["language_feature"],
                        scales=d["scales"], rotations=d["rotations"])
        return torch.autograd.grad([c, f], plist, [dC, dF])

    def step_single():
        out = []
        for v in range(V):
            c, f, r = singles[v](d["means3D"], torch.zeros(0), d["opacities"], shs=d["shs"],
                                 language_feature_precomp=d["language_feature"], scales=d["scales"], rotations=d["rotations"])
            out.append(torch.autograd.grad([c, f], plist, [dC[v], dF[v]]))
        return out

    res = {}
    for name, fn in (("batch", step_batch), ("per-view calls", step_single)):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        n = 30
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        res[name] = (time.perf_counter() - t0) / n * 1e3
    print(f"V={V:2d}: batch {res['batch']:.3f} ms/step = {res['batch'] / V:.3f} ms/view = {P * V / res['batch'] / 1e3:.0f} M Gaussians/s | "
          f"per-view calls {res['per-view calls'] / V:.3f} ms/view", flush=True)
    if os.environ.get("BV_STAGES"):
        _lib.profile_read(True); _lib.set_option("profile", 2)
        for _ in range(10): step_batch()
        torch.cuda.synchronize(); _lib.set_option("profile", 0)
        print("   stages us:", {k: round(ms / max(c, 1) * 1e3) for k, (ms, c) in _lib.profile_read(True).items() if c})
