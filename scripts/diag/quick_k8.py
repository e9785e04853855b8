"""A/B of per-call options on the render kernels' stage times (hipEvent stage timers): python scripts/diag/quick_k8.py key v0 v1 ..."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from manigaussian_amd import GaussianRasterizationSettings, GaussianRasterizer, _lib, check_status
from manigaussian_amd import synthetic as syn

dev = torch.device("cuda:0")
key, vals = sys.argv[1], [int(v) for v in sys.argv[2:]]
for P, F in ((100000, 32), (16384, 3), (100000, 3)):
    W = 128
    sc = {k: v.to(dev).requires_grad_(True) for k, v in syn.make_scene(P, F=F, M=4, seed=0).items()}
    cam = syn.circle_cameras(8, W, W, negative_focal=True)[0]
    rast = GaussianRasterizer(GaussianRasterizationSettings(**syn.camera_settings_kwargs(cam, 1, True, device=dev)))
    dC, dF = [t.to(dev) for t in syn.make_cotangents(W, W, F)]
    m2 = torch.zeros(P, 3, device=dev)

    def step():
        c, f, r = rast(sc["means3D"], m2, sc["opacities"], shs=sc["shs"], language_feature_precomp=sc["language_feature"],
                       scales=sc["scales"], rotations=sc["rotations"])
        return torch.autograd.grad([c, f], list(sc.values()), [dC, dF])

    torch.autograd.set_multithreading_enabled(False)
    for rep in range(2):
        for v in vals:
            _lib.set_option(key, v)
            for _ in range(30):
                step()
            torch.cuda.synchronize()
            _lib.profile_read(reset=True)
            _lib.set_option("profile", 2)
            for _ in range(100):
                step()
            torch.cuda.synchronize()
            _lib.set_option("profile", 0)
            prof = _lib.profile_read(reset=True)
            print(f"P={P} F={F} {key}={v}: " + "  ".join(f"{k} {ms / max(c, 1) * 1e3:.1f}" for k, (ms, c) in prof.items() if c))
    check_status(dev)
_lib.set_option(key, vals[0])
