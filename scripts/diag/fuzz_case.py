"""One case of tests/tools/fuzz_parity.py against the reference's kernels, Gaussian by Gaussian: python scripts/diag/fuzz_case.py N seed index [bin_mode]"""
import os, random, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "tools"))
import util, fuzz_parity as fz
from manigaussian_amd import _lib
n, seed, idx = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
rng = random.Random(seed)
for i in range(n):
    case, op_scale = fz.draw(rng)
    if i == idx:
        break
print(case, op_scale)
sc, cam, kw, dC, dF = util.scene_case(**case)
sc["opacities"] = (sc["opacities"] * op_scale).clamp(max=0.999).contiguous()
inc = case.get("include_feature", True)
for bm in ([int(sys.argv[4])] if len(sys.argv) > 4 else [2, 1, 0]):
    _lib.set_option("bin_mode", bm)
    for rep in range(3):
        ch, fh, rh, gh = util.run_hip(sc, cam, dC, dF, case.get("sh_degree", 1), inc, case["bg"])
        cr, fr, rr, gr, R = util.run_reference(sc, kw, dC, dF)
        line = [f"bin_mode {bm} rep {rep}: radii equal {bool(torch.equal(rh, rr))} color max err {float((ch - cr).abs().max()):.2e}"]
        for k, v in gh.items():
            ref = gr[util.GRAD_KEYS[k]].reshape(v.shape)
            if ref.numel() == 0:
                continue
            e = (v - ref).abs().reshape(v.shape[0], -1).max(1)[0]
            j = int(e.argmax())
            line.append(f"{k}: max err {float(e.max()):.2e} of {float(ref.abs().max()):.2e} at Gaussian {j} (own {float(ref.reshape(v.shape[0], -1)[j].abs().max()):.2e})")
        print(" | ".join(line))
# the reference against itself (its float atomics): two runs
a = util.run_reference(sc, kw, dC, dF)[3]; b = util.run_reference(sc, kw, dC, dF)[3]
print("reference vs reference:", {k: f"{float((a[k] - b[k]).abs().max()):.2e}" for k in a if a[k].numel()})
# which switch removes the difference?  (tight_bins: the {alpha >= 1/255} bbox of the binning; exact_cull: the forward's block cull)
cr, fr, rr, gr, R = util.run_reference(sc, kw, dC, dF)
for opts in (dict(tight_bins=0), dict(exact_cull=0), dict(tight_bins=0, exact_cull=0), dict(gm_waves=16), dict()):
    _lib.set_option("bin_mode", 2)
    old = {k: _lib.get_option(k) for k in opts}
    for k, v in opts.items():
        _lib.set_option(k, v)
    ch, fh, rh, gh = util.run_hip(sc, cam, dC, dF, case.get("sh_degree", 1), inc, case["bg"])
    for k, v in old.items():
        _lib.set_option(k, v)
    d = (ch - cr).abs().max(0)[0]
    y, x = divmod(int(d.argmax()), d.shape[1])
    e = (gh["opacities"] - gr[util.GRAD_KEYS["opacities"]].reshape(gh["opacities"].shape)).abs()
    print(opts, f"color max err {float(d.max()):.2e} at pixel ({x}, {y}); opacity grad max err {float(e.max()):.2e} at Gaussian {int(e.argmax())}")
# Oracle B's view of the pixel / the Gaussian: is a pair there within 2e-5 of a hard threshold?
co, fo, ro, go, st = util.run_oracle_b(sc, kw, dC, dF)
ch, fh, rh, gh = util.run_hip(sc, cam, dC, dF, case.get("sh_degree", 1), inc, case["bg"])
print("HIP vs Oracle B image (robust, fragile, fraction fragile):", util.image_errors(ch, co, st))
print("reference vs Oracle B image:", util.image_errors(cr, co, st))
errs, _ = util.grad_errors_split(gh, go, st)
print("HIP vs Oracle B grads:", {k: tuple(f"{x:.2e}" for x in v) for k, v in errs.items()})
# per-Gaussian geometry, bit for bit, against the reference kernels' GeometryState (tests/test_gpu_parity.py)
import ctypes, types
import numpy as np
from oracle import ref_cuda
from manigaussian_amd import _C
from manigaussian_amd import synthetic as syn
stt = types.SimpleNamespace(**kw)
kwargs = dict(scales=sc["scales"], rotations=sc["rotations"]) if "cov3D_precomp" not in sc else dict(cov3D_precomp=sc["cov3D_precomp"])
ref = ref_cuda.forward_geometry(sc["means3D"], sc["opacities"], stt, shs=sc.get("shs"), language_feature=sc.get("language_feature"), **kwargs)
dev = torch.device("cuda:0")
kwd = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in kw.items()}
d = {k: v.to(dev) for k, v in sc.items()}
e = torch.Tensor([])
P, W, H = case["P"], case["W"], case["H"]
M = sc["shs"].shape[1] if "shs" in sc else 0
out = _C.rasterize_gaussians(kwd["bg"], d["means3D"], e, d["language_feature"], d["opacities"], d["scales"], d["rotations"],
                             1.0, e, kwd["viewmatrix"], kwd["projmatrix"], kwd["tanfovx"], kwd["tanfovy"], H, W, d["shs"], case.get("sh_degree", 1),
                             kwd["campos"], False, False, True)
radii, geom = out[3].cpu().numpy(), out[4].cpu().numpy()
offs = [ctypes.c_size_t(0) for _ in range(4)]
_lib.check(_lib.lib().mgs_debug_geom_layout(P, M, W, H, *[ctypes.byref(o) for o in offs]), "geom layout")
f = lambda o, n: np.frombuffer(geom, np.float32, n, int(o.value)).copy()
rec = f(offs[1], 8 * P).reshape(P, 8)
hip = dict(depths=f(offs[0], P), means2D=rec[:, 0:2], conic_opacity=rec[:, [2, 3, 4, 5]], rgb=f(offs[2], 3 * P).reshape(P, 3))
vis = ref["radii"] > 0
bits = lambda a: np.ascontiguousarray(a).view(np.int32)
for name in ("depths", "means2D", "conic_opacity", "rgb"):
    r_, h_ = ref[name], hip[name]
    dif = (bits(r_) != bits(h_)).reshape(P, -1).any(1) & vis
    print(name, "Gaussians whose bits differ:", np.nonzero(dif)[0][:10], "of", int(vis.sum()), "visible")
g = 56
print("Gaussian 56: xy", hip["means2D"][g], "conic_opacity", hip["conic_opacity"][g], "ref", ref["means2D"][g], ref["conic_opacity"][g])
# the pair (pixel (77, 5), Gaussian 56) in float32 with the kernels' rounding sequence
f32 = np.float32
def fma(a, b, c): return f32(np.float64(a) * np.float64(b) + np.float64(c))
x, y = hip["means2D"][g]; cx, cy, cz, o = hip["conic_opacity"][g]
for px in (76, 77, 78):
    dx, dy = f32(x - f32(px)), f32(y - f32(5))
    t = fma(f32(cx * dx), dx, f32(f32(cz * dy) * dy))
    p = fma(f32(-0.5), t, -f32(f32(cy * dx) * dy))
    G = np.exp(np.float64(p)); a = np.float64(o) * G
    print(f"pixel ({px}, 5): power {p!r} alpha ~ {a:.9f} (1/255 = {1/255:.9f}, diff {(a - 1/255):.3e})")
import subprocess
hexs = [format(int(np.float32(v).view(np.uint32)), "08x") for v in (x, y, cx, cy, cz, o, 77.0, 5.0)]
print("pair_probe args:", " ".join(hexs))
if subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", os.path.join(ROOT, "scripts/diag/pair_probe.hip"), "-o", "/tmp/pair_probe"]).returncode == 0:
    print(subprocess.run(["/tmp/pair_probe"] + hexs, capture_output=True, text=True).stdout)
