"""How far is the forward preprocess from the reference's, bit for bit?  Per shape: the Gaussians (radii > 0 on both sides)
whose depth / pixel mean / conic / opacity / rgb / cov3D differ from the reference kernels' GeometryState, and by how many ulps."""
import ctypes, os, sys, types
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import util
from manigaussian_amd import _C, _lib
from manigaussian_amd import synthetic as syn
from oracle import ref_cuda

def ulps(a, b):
    ia, ib = a.view(np.int32).astype(np.int64), b.view(np.int32).astype(np.int64)
    return np.abs(ia - ib)

def hip_geometry(sc, cam, case):
    dev = torch.device("cuda:0")
    kwd = syn.camera_settings_kwargs(cam, case.get("sh_degree", 1), True, bg=case.get("bg", (0.1, 0.2, 0.3)), device=dev)
    d = {k: v.to(dev) for k, v in sc.items()}
    e = torch.Tensor([])
    P, M, W, H = d["means3D"].shape[0], d["shs"].shape[1], kwd["image_width"], kwd["image_height"]
    out = _C.rasterize_gaussians(kwd["bg"], d["means3D"], e, d["language_feature"], d["opacities"], d["scales"], d["rotations"],
                                 1.0, e, kwd["viewmatrix"], kwd["projmatrix"], kwd["tanfovx"], kwd["tanfovy"], H, W, d["shs"],
                                 case.get("sh_degree", 1), kwd["campos"], False, False, True)
    radii, geom = out[3].cpu().numpy(), out[4].cpu().numpy()
    offs = [ctypes.c_size_t(0) for _ in range(4)]
    _lib.check(_lib.lib().mgs_debug_geom_layout(P, M, W, H, *[ctypes.byref(o) for o in offs]), "geom layout")
    f = lambda o, n: np.frombuffer(geom, np.float32, n, int(o.value)).copy()
    rec = f(offs[1], 8 * P).reshape(P, 8)
    return dict(radii=radii, depths=f(offs[0], P), means2D=rec[:, 0:2], conic_opacity=np.stack([rec[:, 2], rec[:, 3], rec[:, 4], rec[:, 5]], 1),
                rgb=f(offs[2], 3 * P).reshape(P, 3), cov3D=f(offs[3], 6 * P).reshape(P, 6))

