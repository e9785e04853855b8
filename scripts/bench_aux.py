"""Timings of the 8f-row kernels against the reference's torch ops at ManiGaussian's sizes (N = 16 384 points)."""
import math, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from manigaussian_amd.regressor import gaussian_epilogue
from manigaussian_amd.voxel import point_latent_pe

dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)


def timeit(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6


N, C, G = 16384, 128, 100
bounds = (-0.3, -0.5, 0.6, 0.7, 0.5, 1.6)
vox = torch.randn(1, C, G, G, G, generator=g).to(dev).requires_grad_(True)
lo, hi = torch.tensor(bounds[:3], device=dev), torch.tensor(bounds[3:], device=dev)
xyz = (lo + (hi - lo) * torch.rand(1, N, 3, generator=g).to(dev))
w = torch.randn(N, C + 39, generator=g).to(dev)
freqs = torch.repeat_interleave(math.pi * 2.0 ** torch.arange(0, 6), 2).view(1, -1, 1).to(dev)
phases = torch.zeros(12, device=dev); phases[1::2] = math.pi * 0.5


def ref_latent():
    canon = (xyz - lo) / (hi - lo)
    grid = (canon * 2 - 1.0).unsqueeze(1).unsqueeze(1)
    pl = torch.nn.functional.grid_sample(vox, grid, align_corners=True, mode="bilinear").squeeze(2).squeeze(2).permute(0, 2, 1)
    x = canon.reshape(-1, 3)
    emb = torch.sin(torch.addcmul(phases.view(1, -1, 1), x.unsqueeze(1).repeat(1, 12, 1), freqs)).view(N, -1)
    return torch.cat((pl.reshape(-1, C), torch.cat((x, emb), dim=-1)), dim=-1)


def fb(fn):
    def run():
        vox.grad = None
        (fn() * w).sum().backward()
    return run


print(f"point latent + PE, N={N}, volume {C}x{G}^3: fwd  hip {timeit(lambda: point_latent_pe(vox, xyz, bounds)):.0f} us | torch {timeit(ref_latent):.0f} us")
print(f"                                           fwd+bwd hip {timeit(fb(lambda: point_latent_pe(vox, xyz, bounds))):.0f} us | torch {timeit(fb(ref_latent)):.0f} us")

raw = torch.randn(1, N, 26, generator=g).to(dev).requires_grad_(True)
xin = torch.randn(1, N, 3, generator=g).to(dev)


def ref_epi():
    dxyz, op, sc, rt, fdc, feat, frest = raw.split([3, 1, 3, 4, 3, 3, 9], dim=-1)
    sh = torch.cat([fdc.unsqueeze(2), frest.reshape(1, N, -1, 3)], dim=2)
    return (xin + dxyz, torch.sigmoid(op), torch.clamp_max(torch.exp(sc), 0.05), torch.nn.functional.normalize(rt, dim=-1), sh,
            feat / (feat.norm(dim=-1, keepdim=True) + 1e-12))


def hip_epi():
    o = gaussian_epilogue(raw, xin)
    return o["xyz"], o["opacity"], o["scale"], o["rot"], o["sh"], o["feature_normalized"]


def fb2(fn):
    def run():
        raw.grad = None
        sum(t.sum() for t in fn()).backward()
    return run


print(f"regressor epilogue, N={N}: fwd hip {timeit(hip_epi):.0f} us | torch {timeit(ref_epi):.0f} us ; fwd+bwd hip {timeit(fb2(hip_epi)):.0f} us | torch {timeit(fb2(ref_epi)):.0f} us")

# --- camera calibration: the reference's per-step host round trip vs one kernel / a cache hit
import math  # noqa: E402
import time  # noqa: E402

import numpy as np  # noqa: E402

from manigaussian_amd import camera, synthetic as syn  # noqa: E402

bs, Wc, Hc = 2, 128, 128   # update() calibrates the current and the next frame's camera
c2w = np.stack([syn.look_at_c2w((1.5, 0.3 * i, 1.8), (0.2, 0.0, 0.9), flip_xy=True) for i in range(bs)])
Kc = np.stack([np.array([[-175.8, 0, 64.0], [0, -175.8, 64.0], [0, 0, 1.0]])] * bs)
data = {"intr": torch.from_numpy(Kc).float().to(dev), "extr": torch.from_numpy(c2w).float().to(dev)}


def ref_calib():
    """NeuralRenderer.get_novel_calib's data flow (neural_rendering.py:217-248) with the restated math, followed by the
    four scalar read-backs render() does per view (gaussian_renderer/__init__.py:35-39)."""
    cams = [syn.novel_calib(data["extr"][i].cpu().numpy(), data["intr"][i].cpu().numpy().astype(np.float64), Wc, Hc)
            for i in range(bs)]
    nv = {"FovX": torch.FloatTensor(np.array([c["FovX"] for c in cams])).to(dev),
          "FovY": torch.FloatTensor(np.array([c["FovY"] for c in cams])).to(dev),
          "width": torch.tensor([Wc] * bs).to(dev), "height": torch.tensor([Hc] * bs).to(dev),
          "world_view_transform": torch.stack([c["world_view_transform"] for c in cams]).to(dev),
          "full_proj_transform": torch.stack([c["full_proj_transform"] for c in cams]).to(dev),
          "camera_center": torch.stack([c["camera_center"] for c in cams]).to(dev)}
    for i in range(bs):
        math.tan(nv["FovX"][i] * 0.5), math.tan(nv["FovY"][i] * 0.5), int(nv["height"][i]), int(nv["width"][i])
    return nv


def host_time(fn, n=200):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


print(f"camera calibration, {bs} cameras per step (host wall time incl. synchronisations): reference flow "
      f"{host_time(ref_calib):.0f} us | device kernel (no sync) {host_time(lambda: camera.get_novel_calib(data, Wc, Hc)):.0f} us")
