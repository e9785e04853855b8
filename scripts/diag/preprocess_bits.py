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

for case in (dict(P=100000, F=32, W=128, H=128, neg=True, bg=(0.0, 0.0, 0.0), seed=0, cam_index=0),
             dict(P=500000, F=32, W=256, H=256, neg=True, bg=(0.0, 0.0, 0.0), seed=0),
             dict(P=100000, F=32, W=128, H=128, neg=False, bg=(0.0, 0.0, 0.0), seed=5),
             dict(P=100000, F=32, W=128, H=128, neg=True, bg=(0.0, 0.0, 0.0), seed=0, phase=0.37),
             dict(P=100000, F=32, W=128, H=128, neg=False, bg=(0.0, 0.0, 0.0), seed=7, phase=1.1)):
    phase = case.pop("phase", 0.0)
    sc, cam, kw, dC, dF = util.scene_case(**case)
    if phase:  # a camera without structural zeros in its matrices (the test cameras sit at multiples of 90 degrees)
        cam = syn.circle_cameras(4, case["W"], case["H"], negative_focal=case["neg"], phase=phase)[1]
        kw = syn.camera_settings_kwargs(cam, 1, True, bg=case["bg"])
    st = types.SimpleNamespace(**kw)
    ref = ref_cuda.forward_geometry(sc["means3D"], sc["opacities"], st, shs=sc["shs"], language_feature=sc["language_feature"],
                                    scales=sc["scales"], rotations=sc["rotations"])
    hip = hip_geometry(sc, cam, case)
    vis = (ref["radii"] > 0) & (hip["radii"] > 0)
    print(f"P={case['P']} {case['W']}x{case['H']} neg={case['neg']} phase={phase}: radii equal {np.array_equal(ref['radii'], hip['radii'])}, visible {int(vis.sum())}")
    for name in ("depths", "means2D", "conic_opacity", "rgb", "cov3D"):
        a, b = ref[name][vis], hip[name][vis]
        u = ulps(np.ascontiguousarray(a), np.ascontiguousarray(b)).reshape(a.shape[0], -1)
        rows = (u > 0).any(1)
        print(f"   {name:14s} Gaussians that differ: {int(rows.sum()):7d}  max ulps {int(u.max())}  per column {[(int((u[:, c] > 0).sum()), int(u[:, c].max())) for c in range(u.shape[1])]}")

# ---- a closer look at cov3D: a few Gaussians that differ, both sides and the float64 value
case = dict(P=100000, F=32, W=128, H=128, neg=True, bg=(0.0, 0.0, 0.0), seed=0, cam_index=0)
sc, cam, kw, dC, dF = util.scene_case(**case)
st = types.SimpleNamespace(**kw)
ref = ref_cuda.forward_geometry(sc["means3D"], sc["opacities"], st, shs=sc["shs"], language_feature=sc["language_feature"],
                                scales=sc["scales"], rotations=sc["rotations"])
hip = hip_geometry(sc, cam, case)
vis = (ref["radii"] > 0) & (hip["radii"] > 0)
u = ulps(np.ascontiguousarray(ref["cov3D"]), np.ascontiguousarray(hip["cov3D"]))
idx = np.where(vis & (u[:, 0] > 3))[0][:4]
for i in idx:
    s = sc["scales"][i].double().numpy(); q = sc["rotations"][i].double().numpy()
    r, x, y, z = q
    R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y)],
                  [2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x)],
                  [2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)]])  # R[i][k] = glm column i, component k
    M = R * s[None, :]
    Sig = M @ M.T
    exact = np.array([Sig[0, 0], Sig[0, 1], Sig[0, 2], Sig[1, 1], Sig[1, 2], Sig[2, 2]])
    # float32 inputs, exact algebra on them: the "true" value both sides approximate
    print(i, "scale", sc["scales"][i].tolist(), "rot", sc["rotations"][i].tolist())
    print("   ref   ", ref["cov3D"][i].tolist())
    print("   hip   ", hip["cov3D"][i].tolist())
    print("   f64   ", exact.tolist())
    print("   ulps ref-f64", [int(abs(np.float32(e).view(np.int32).astype(np.int64) - np.float32(a).view(np.int32).astype(np.int64))) for e, a in zip(exact, ref["cov3D"][i])],
          "hip-f64", [int(abs(np.float32(e).view(np.int32).astype(np.int64) - np.float32(a).view(np.int32).astype(np.int64))) for e, a in zip(exact, hip["cov3D"][i])])
