"""The reference's own kernels (oracle/_ref, built with hipcc) timed on this MI355X beside the HIP path, same workload.

  python tests/tools/ref_bench.py [P=100000] [size=128] [steps=200]

Workloads: BASELINE configs[2] (32 feature channels; the reference rebuilt at that width) and configs[1] (3 channels,
the stock reference build).  Both sides keep inputs resident and include their per-call zero-fills; the reference side
runs without PyTorch (ref_wrapper.cu:ref_bench: its forward has the num_rendered read-back of rasterizer_impl.cu:282),
this side runs through the full Python/autograd path exactly like bench.py.  Test infrastructure, not product.
"""
import json
import os
import sys
import time
import types

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from manigaussian_amd import GaussianRasterizationSettings, GaussianRasterizer  # noqa: E402
from manigaussian_amd import synthetic as syn  # noqa: E402
from oracle import ref_cuda  # noqa: E402


def ours(sc, cam, dC, dF, steps, warmup=20):
    dev = torch.device("cuda:0")
    params = {k: v.to(dev).requires_grad_(True) for k, v in sc.items()}
    P = sc["means3D"].shape[0]
    means2D = torch.zeros(P, 3, device=dev, requires_grad=True)
    rast = GaussianRasterizer(GaussianRasterizationSettings(**syn.camera_settings_kwargs(cam, 1, True, device=dev)))
    dC, dF = dC.to(dev), dF.to(dev)
    plist = list(params.values())
    torch.autograd.set_multithreading_enabled(False)

    def step():
        color, feat, radii = rast(means3D=params["means3D"], means2D=means2D, opacities=params["opacities"],
                                  shs=params["shs"], language_feature_precomp=params["language_feature"],
                                  scales=params["scales"], rotations=params["rotations"])
        return torch.autograd.grad([color, feat], plist, [dC, dF])
    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


def main():
    kv = dict(a.split("=") for a in sys.argv[1:])
    P, size, steps = int(kv.get("P", 100000)), int(kv.get("size", 128)), int(kv.get("steps", 200))
    out = []
    for F in (32, 3):
        if not ref_cuda.available(F):
            print(f"no reference build for F={F}", file=sys.stderr)
            continue
        sc = syn.make_scene(P, F=F, M=4, seed=0)
        cam = syn.circle_cameras(8, size, size, negative_focal=True)[0]
        dC, dF = syn.make_cotangents(size, size, F, seed=1)
        st = types.SimpleNamespace(**syn.camera_settings_kwargs(cam, 1, True))
        r = ref_cuda.bench(sc["means3D"], sc["opacities"], st, dC, dF, warmup=10, iters=steps, shs=sc["shs"],
                           language_feature=sc["language_feature"], scales=sc["scales"], rotations=sc["rotations"])
        ms = ours(sc, cam, dC, dF, steps)
        rec = dict(workload=f"{P} Gaussians, {size}x{size}, RGB SH deg 1 + {F} feature ch, fwd+bwd", F=F,
                   reference_ms_step=r["ms_step"], reference_ms_fwd=r["ms_fwd"], reference_ms_bwd=r["ms_bwd"],
                   reference_gaussians_per_s=P / r["ms_step"] * 1e3, num_rendered_reference=r["num_rendered"],
                   hip_ms_step=ms, hip_gaussians_per_s=P / ms * 1e3, speedup=r["ms_step"] / ms)
        out.append(rec)
        print(json.dumps(rec), flush=True)


if __name__ == "__main__":
    main()
