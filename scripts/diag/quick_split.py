"""A/B of the bucket rank's parts per tile (MgsOptions.dbg bits 12-14) on the binning stage time: python scripts/diag/quick_split.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from manigaussian_amd import GaussianRasterizationSettings, GaussianRasterizer, _lib, check_status
from manigaussian_amd import synthetic as syn

dev = torch.device("cuda:0")
for P, F, W in ((500000, 32, 256), (100000, 32, 128), (16384, 3, 128), (200000, 3, 512)):
    sc = {k: v.to(dev) for k, v in syn.make_scene(P, F=F, M=4, seed=0).items()}
    cam = syn.circle_cameras(8, W, W, negative_focal=True)[0]
    rast = GaussianRasterizer(GaussianRasterizationSettings(**syn.camera_settings_kwargs(cam, 1, True, device=dev)))
    m2 = torch.zeros(P, 3, device=dev)

    def step():
        with torch.no_grad():
            return rast(sc["means3D"], m2, sc["opacities"], shs=sc["shs"], language_feature_precomp=sc["language_feature"],
                        scales=sc["scales"], rotations=sc["rotations"])

    for code in (0, 1, 2, 3, 4):
        _lib.set_option("dbg", code << 12)
        for _ in range(10):
            step()
        torch.cuda.synchronize()
        _lib.profile_read(reset=True)
        _lib.set_option("profile", 2)
        for _ in range(50):
            step()
        torch.cuda.synchronize()
        _lib.set_option("profile", 0)
        prof = _lib.profile_read(reset=True)
        ms, c = prof["bin_segsort"]
        print(f"P={P} {W}x{W}: parts per tile {'auto' if code == 0 else 1 << (code - 1)}: bucket rank {ms / max(c, 1) * 1e3:.1f} us (hipEvent pair)")
    check_status(dev)
_lib.set_option("dbg", 0)
