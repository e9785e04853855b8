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
