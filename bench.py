#!/usr/bin/env python
"""bench.py -- Gaussians rasterized/sec (fwd+bwd) on MI355X, BASELINE.json's metric.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: launched by `python -m torch.distributed.run --nproc-per-node N ...`, one rank per GPU.)

Workload at N=1 = BASELINE.json configs[2] ("100k Gaussians, 128x128, RGB + 32-ch language feature map,
1xMI355X"), the configuration the metric string is quoted on (128x128, 32 feat-ch): P=100 000 synthetic
Gaussians (SURVEY.md 8d statistics), SH degree 1 (M=4), F=32, one look-at view per GPU per step, production
negative-focal cameras, inputs resident in HBM.  A step = one forward + one backward of the rasterizer
through the public GaussianRasterizer autograd API (+ one all-reduce of the flat per-Gaussian gradient
buffer when N>1; weak scaling: every GPU renders its own view of the replicated Gaussian set).

The JSON line also carries
  roofline:     the dominant kernel (render backward) timed live with HIP events on its launch stream,
                achieved = algorithmic bytes (R*(112+12F) + N_pix*(20+4F)) / mean duration,
                against the 8 TB/s HBM peak; traffic = measured HBM bytes per launch from the committed
                rocprofv3 PMC passes of this command (profiles/pmc_traffic.json).
  cpu_baseline: Oracle B (oracle/mgs_oracle.c, a port: the reference has no CPU rasterizer) on the host
                cores, same workload, a bounded number of fwd+bwd passes.
"""
import argparse
import json
import os
import sys
import time
import types

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from manigaussian_amd import GaussianRasterizationSettings, GaussianRasterizer, _lib  # noqa: E402
from manigaussian_amd import synthetic as syn  # noqa: E402
from manigaussian_amd.parallel import all_reduce_grads  # noqa: E402

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--P", type=int, default=100000)
    ap.add_argument("--F", type=int, default=32)
    ap.add_argument("--size", type=int, default=128)
    ap.add_argument("--views", type=int, default=1, help="views of the Gaussian set each GPU renders per step (> 1: one "
                    "batched call, SURVEY 8f row 1; the headline config is 1)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--tight-bins", type=int, default=None)
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only to exercise "
                                                      "the N>1 path on a box with fewer GPUs than ranks)")
    ap.add_argument("--one-device", action="store_true", help="testing only: every rank uses cuda:0")
    return ap.parse_args()


def cpu_baseline(sc, cam, d_color, d_feat, P, max_seconds):
    """Oracle B fwd+bwd on the host cores (the checker, timed beside the GPU path; never the product)."""
    from oracle import oracle_b
    oracle_b.build()
    st = types.SimpleNamespace(**syn.camera_settings_kwargs(cam, 1, True))
    cores = oracle_b.max_threads()
    times = []
    t_start = time.perf_counter()
    while True:
        t0 = time.perf_counter()
        _, _, _, state = oracle_b.forward(sc["means3D"], sc["opacities"], st, shs=sc["shs"],
                                          language_feature=sc["language_feature"], scales=sc["scales"],
                                          rotations=sc["rotations"])
        oracle_b.backward(state, d_color, d_feat)
        times.append(time.perf_counter() - t0)
        del state
        if len(times) >= 5 or time.perf_counter() - t_start > max_seconds:
            break
    best = min(times)
    return {"value": P / best, "unit": "Gaussians/s", "cores": cores, "kind": "port",
            "sample": f"{len(times)} fwd+bwd passes of the same workload (1 view), best of {len(times)}: "
                      f"{best * 1e3:.1f} ms, OpenMP {cores} threads"}


def pmc_traffic(kernel):
    """HBM bytes per launch of `kernel` from the committed PMC passes of this same command
    (profiles/pmc_traffic.json, written by scripts/pmc_to_json.py; rocprofv3 cannot run inside the bench)."""
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
            for k, v in json.load(f)["kernels"].items():
                if kernel in k:
                    return v["hbm_bytes_per_launch"]
    except (OSError, KeyError, ValueError):
        pass
    return None


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (there is no CPU fallback for the product path)")
    if args.one_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(args.backend)
    n_gpus = world
    if args.tight_bins is not None:
        _lib.set_option("tight_bins", args.tight_bins)

    P, F, W, H = args.P, args.F, args.size, args.size
    sc = syn.make_scene(P, F=F, M=4, seed=0)  # identical on every rank: the replicated Gaussian set
    cams = syn.circle_cameras(max(n_gpus, 8), W, H, negative_focal=True)
    V = max(1, args.views)
    my_cams = [cams[(rank * V + i) % len(cams)] for i in range(V)]
    cam = my_cams[0]
    d_color_h, d_feat_h = syn.make_cotangents(W, H, F, seed=1 + rank)
    params = {k: v.to(dev).requires_grad_(True) for k, v in sc.items()}
    means2D = torch.zeros(P, 3, device=dev, requires_grad=True)
    d_color, d_feat = d_color_h.to(dev), d_feat_h.to(dev)
    all_settings = [GaussianRasterizationSettings(**syn.camera_settings_kwargs(c, 1, True, device=dev)) for c in my_cams]
    settings = all_settings[0]
    rast = GaussianRasterizer(settings)
    if V > 1:  # one batched call per step: V views of the same Gaussian set, gradients summed over the views on the device
        from manigaussian_amd import GaussianRasterizerBatch
        rast_batch = GaussianRasterizerBatch(all_settings)
        d_color = torch.stack([syn.make_cotangents(W, H, F, seed=1 + rank * V + i)[0] for i in range(V)]).to(dev)
        d_feat = torch.stack([syn.make_cotangents(W, H, F, seed=1 + rank * V + i)[1] for i in range(V)]).to(dev)
    plist = list(params.values())
    pending = []  # (work handle, gradients) of the all-reduce still in flight

    def drain():
        while pending:
            h, _ = pending.pop()
            if h is not None:
                h.wait()

    def step():
        if V > 1:
            color, feat, radii = rast_batch(params["means3D"], None, params["opacities"], shs=params["shs"],
                                            language_feature_precomp=params["language_feature"],
                                            scales=params["scales"], rotations=params["rotations"])
        else:
            color, feat, radii = rast(means3D=params["means3D"], means2D=means2D, opacities=params["opacities"],
                                      shs=params["shs"], language_feature_precomp=params["language_feature"],
                                      scales=params["scales"], rotations=params["rotations"])
        grads = torch.autograd.grad([color, feat], plist, [d_color, d_feat])
        # N > 1: ONE in-place all-reduce of the allocation all gradients alias, asynchronous on RCCL's stream: it
        # overlaps the next step's forward/backward (which write a fresh allocation); a step's reduced gradients are
        # complete when the following step issues its own all-reduce (a trainer's optimizer would wait right there).
        drain()
        pending.append((all_reduce_grads(grads, async_op=True), grads))
        return grads

    def sync_all():
        drain()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    # Autograd scheduling knob, results unchanged: run the backward on the calling thread instead of handing it
    # to the engine's device thread (the hand-off costs ~50-100 us of host time per step on this node, which at
    # ~0.25 ms per step is not noise).  DESIGN.md section 8.
    torch.autograd.set_multithreading_enabled(False)
    for _ in range(args.warmup):
        step()
    sync_all()
    _lib.profile_read(reset=True)
    _lib.set_option("profile", 1)  # two hipEvents per step around the render backward only
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync_all()
    elapsed = time.perf_counter() - t0
    _lib.set_option("profile", 0)
    prof = _lib.profile_read(reset=True)
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # stage breakdown, untimed extra pass
    _lib.set_option("profile", 2)
    n_extra = min(args.steps, 20)
    for _ in range(n_extra):
        step()
    drain()
    torch.cuda.synchronize()
    _lib.set_option("profile", 0)
    stages = {k: (ms / max(c, 1)) for k, (ms, c) in _lib.profile_read(reset=True).items()}

    # measured instance count (R) of this rank's view(s): what one launch of the render kernels processes
    from manigaussian_amd import _C
    with torch.no_grad():
        e = torch.empty(0, device=dev)
        R = sum(_C.rasterize_gaussians(st.bg, params["means3D"], e, params["language_feature"], params["opacities"],
                                       params["scales"], params["rotations"], 1.0, e, st.viewmatrix, st.projmatrix,
                                       st.tanfovx, st.tanfovy, H, W, params["shs"], 1, st.campos, False, False,
                                       True)[0] for st in all_settings)

    if rank == 0:
        ms_step = elapsed / args.steps * 1e3
        value = P * V * n_gpus * args.steps / elapsed  # V views per GPU per step (headline: 1)
        bwd_ms, bwd_n = prof["render_bwd"]
        bwd_avg_ms = bwd_ms / max(bwd_n, 1)
        npix = W * H * V  # pixels one launch covers
        bytes_k8 = R * (112 + 12 * F) + npix * (20 + 4 * F)
        achieved = bytes_k8 / (bwd_avg_ms * 1e-3) / 1e9 if bwd_avg_ms > 0 else 0.0
        M = 4
        bytes_path = V * P * (434 + 48 * M + 4 * F) + R * (196 + 16 * F) + npix * (40 + 8 * F)  # all V views
        out = {
            "metric": "Gaussians rasterized/sec (fwd+bwd), 128x128, 32 feat-ch; HBM GB/s vs peak",
            "value": value, "unit": "Gaussians/s", "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"configs[2]: {P} Gaussians, {W}x{H}, RGB via SH deg 1 (M=4) + {F}-ch language "
                                   f"feature, fwd+bwd, {V} view{'s (one batched call)' if V > 1 else ''} per GPU per step, "
                                   f"negative-focal look-at cameras",
                       "P": P, "W": W, "H": H, "F": F, "M": M, "views_per_gpu": V, "num_rendered_R": int(R),
                       "R_over_P": R / (P * V), "tight_bins": _lib.get_option("tight_bins"),
                       "collective": "1 in-place all-reduce of the flat per-Gaussian gradient buffer" if n_gpus > 1 else "none"},
            "roofline": {"bound": "hbm", "kernel": "K8 render backward (gm_bwd_kernel)", "achieved": achieved,
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": pmc_traffic("gm_bwd_kernel"),
                         "algorithmic_bytes_per_launch": bytes_k8, "avg_launch_ms": bwd_avg_ms, "launches": bwd_n},
            "path_hbm": {"algorithmic_bytes_per_step_per_gpu": bytes_path,
                         "achieved_GBps": bytes_path * n_gpus / (ms_step * 1e-3) / 1e9,
                         "frac_of_peak": bytes_path / (ms_step * 1e-3) / 1e9 / HBM_PEAK_GBS},
            "stages_ms": stages,
        }
        if not args.no_cpu_baseline and n_gpus == 1:
            out["cpu_baseline"] = cpu_baseline(sc, cam, d_color_h, d_feat_h, P, args.cpu_seconds)
            out["gpu_over_cpu"] = value / out["cpu_baseline"]["value"]
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
