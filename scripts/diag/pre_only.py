"""Forward preprocess alone, through the two-call C entry point (mgs_rasterize_forward_preprocess: preprocess + count
read-back, nothing downstream): python scripts/diag/pre_only.py [reps].  Run under rocprofv3 --kernel-trace --stats to get the
kernel's duration at BASELINE configs[2] (100 000 Gaussians, 128 x 128) and at the configs[4] shape (500 000, 256 x 256).
Safe with timing-only builds of the library whose reservations are garbage: nothing consumes them here."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from manigaussian_amd import _C, _lib  # noqa: E402
from manigaussian_amd import synthetic as syn  # noqa: E402


def run(P, W, reps):
    dev = torch.device("cuda:0")
    F, M = 32, 4
    sc = {k: v.to(dev) for k, v in syn.make_scene(P, F=F, M=M, seed=0).items()}
    cam = syn.circle_cameras(8, W, W, negative_focal=True)[0]
    kw = syn.camera_settings_kwargs(cam, 1, True, device=dev)
    L = _lib.lib()
    u8 = dict(dtype=torch.uint8, device=dev)
    geom = torch.empty(L.mgs_geom_bytes(P, M, W, W), **u8)
    img = torch.empty(L.mgs_img_bytes(W, W), **u8)
    e = torch.empty(0, device=dev)
    a = _lib.MgsRasterArgs()
    _C._fill_args(a, P=P, D=1, M=M, F=F, W=W, H=W, tanfovx=kw["tanfovx"], tanfovy=kw["tanfovy"], scale_modifier=1.0,
                  prefiltered=False, debug=False, include_feature=True, background=kw["bg"], means3D=sc["means3D"],
                  sh=sc["shs"], colors=e, language_feature=sc["language_feature"], opacity=sc["opacities"],
                  scales=sc["scales"], rotations=sc["rotations"], cov3D_precomp=e, viewmatrix=kw["viewmatrix"],
                  projmatrix=kw["projmatrix"], campos=kw["campos"], geom=geom, binning=None, img=img)
    radii = torch.empty(P, dtype=torch.int32, device=dev)
    nr = ctypes.c_int32(0)
    for _ in range(reps):
        rc = L.mgs_rasterize_forward_preprocess(ctypes.byref(a), radii.data_ptr(), ctypes.byref(nr), None)
        assert rc == 0, _lib.last_error()
    torch.cuda.synchronize()
    print(f"P={P} {W}x{W}: num_rendered {nr.value}, visible {int((radii > 0).sum())}", flush=True)


if __name__ == "__main__":
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    run(100000, 128, reps)
    run(500000, 256, reps)
