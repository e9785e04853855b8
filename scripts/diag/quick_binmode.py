"""Binning stage times, bucket rank (bin_mode 2) against segment sort + rank merge (bin_mode 1), over shapes."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from manigaussian_amd import GaussianRasterizationSettings, GaussianRasterizer, _lib, check_status
from manigaussian_amd import synthetic as syn

dev = torch.device("cuda:0")
for P, F, W in ((200000, 3, 512), (50000, 3, 512), (400000, 3, 1024), (5000, 3, 128), (2000000, 3, 128), (500000, 32, 256)):
    sc = {k: v.to(dev) for k, v in syn.make_scene(P, F=F, M=4, seed=0).items()}
    cam = syn.circle_cameras(8, W, W, negative_focal=True)[0]
    rast = GaussianRasterizer(GaussianRasterizationSettings(**syn.camera_settings_kwargs(cam, 1, True, device=dev)))
    m2 = torch.zeros(P, 3, device=dev)

    def step():
        with torch.no_grad():
            return rast(sc["means3D"], m2, sc["opacities"], shs=sc["shs"], language_feature_precomp=sc["language_feature"],
                        scales=sc["scales"], rotations=sc["rotations"])

    for mode in (2, 1):
        _lib.set_option("bin_mode", mode)
        for _ in range(10):
            step()
        torch.cuda.synchronize()
        _lib.profile_read(reset=True)
        _lib.set_option("profile", 2)
        for _ in range(30):
            step()
        torch.cuda.synchronize()
        _lib.set_option("profile", 0)
        prof = _lib.profile_read(reset=True)
        t = {k: ms / max(c, 1) * 1e3 for k, (ms, c) in prof.items() if k.startswith("bin_") and c}
        print(f"P={P} {W}x{W} bin_mode {mode}: " + "  ".join(f"{k} {v:.1f}" for k, v in t.items()) + f"  sum {sum(t.values()):.1f} us")
    check_status(dev)
_lib.set_option("bin_mode", 2)
