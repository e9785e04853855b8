#!/usr/bin/env python
"""bench.py -- Gaussians rasterized/sec (fwd+bwd) on MI355X, BASELINE.json's metric.

  python bench.py --gpus N --steps K --warmup W [--config c3|c2|c5shape|ref16k|c4|c5] [--mode graph|eager|eager-st]

N > 1 without a launcher: the script re-executes itself under `python -m torch.distributed.run --nproc-per-node N`
(one rank per GPU, RCCL); under a launcher (RANK / WORLD_SIZE set) it checks WORLD_SIZE == N.

Default workload = BASELINE.json configs[2] ("100k Gaussians, 128x128, RGB + 32-ch language feature map, 1xMI355X"), the
configuration the metric string is quoted on: P=100 000 synthetic Gaussians (SURVEY.md 8d statistics), SH degree 1 (M=4),
F=32, one look-at view per GPU per step, production negative-focal cameras, inputs resident in HBM.  A step = one forward
+ one backward of the rasterizer through the public GaussianRasterizer autograd API (+ one all-reduce of the per-Gaussian
parameter gradients when N>1; weak scaling: every GPU renders its own view of the replicated Gaussian set).

Modes (all three are timed and reported under "modes_ms_per_step"; `value` comes from --mode, default graph):
  graph     the step captured once with torch.cuda.graph through the PUBLIC autograd API and replayed (possible because
            nothing in the library synchronises).  The default: it needs no process-wide torch switch and is what a
            training loop that cares about a 0.17 ms step would do; one replay costs ~7 us more than the kernels.
  eager-st  the Python step called K times with torch.autograd.set_multithreading_enabled(False): the backward is
            enqueued by the calling thread, the host runs ahead and the step is GPU-bound: ms_per_step == the sum of the
            kernel durations of rocprofv3 (profiles/).  A process-global switch, hence not the headline.
  eager     the same with torch's default autograd threading: every backward is handed to the engine's device thread and
            back (two thread wake-ups per step): host-bound at this step size, printed beside the headline.

EXACTLY --steps steps are timed for `value` / `ms_per_step` / `steps`.  A K-step region shorter than 50 ms is additionally
re-measured over a longer region and reported as `long_run` (the short region carries ~0.1 ms of bracket overhead).

Dynamic configs (BASELINE configs[3] / [4], strong scaling: the work of a step is fixed, the ranks share it):
  c4  100 000 Gaussians, DeformationField (fp32 MLP 70 -> 512 x 5 -> 7) per timestep, 4 timesteps x 4 views = 16 renders per
      step; rank r takes a contiguous share of the (timestep, view) items (4 GPUs: one timestep x 4 views each).
  c5  500 000 Gaussians, 256 x 256, F = 32, DeformationField, 8 views per step (8 / N views per GPU: 8, 4, 2, 1).
  Per timestep on a rank: input assembly (HIP) -> MLP GEMMs (torch / hipBLASLt) + fused elementwise passes (HIP) -> apply
  (HIP) -> ONE batched render of that rank's views -> l2(rgb) + 0.01 l2(feature) -> backward into the MLP parameters (a flat
  gradient bucket: the one all-reduce when N > 1, asynchronous) and into point_latent (a local leaf, not reduced).

The JSON line also carries
  roofline      the dominant kernel (render backward) timed live with HIP events on its launch stream; what bounds it
                (VALU issue, from the committed counter passes profiles/r02_sq_counters.json, tied to the library's
                hash) and its HBM line: algorithmic bytes (R*(112+12F) + N_pix*(20+4F)) / duration vs
                8 TB/s, counter bytes per launch;
  cpu_baseline  Oracle B (oracle/mgs_oracle.c, a port: the reference has no CPU rasterizer) on the host cores, same
                workload, a bounded number of fwd+bwd passes.
"""
import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time
import types

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0     # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s
VALU_PEAK_GIPS = 1228.8   # 256 CUs x 4 SIMDs x 2.4 GHz / 2 cycles per wave64 VALU instruction (same guide)
MIN_TIMED_MS = 50.0       # a timed region shorter than this is re-measured over a longer one, reported beside it (`long_run`)
EXTRA_WARMUP_MS = 30.0    # untimed steps beyond --warmup until the device has been busy this long (clock ramp)

CONFIGS = {
    "c3": dict(P=100000, F=32, size=128, views=1, renders=1, label="configs[2]"),
    "c2": dict(P=100000, F=3, size=128, views=1, renders=1, label="configs[1]"),
    "c5shape": dict(P=500000, F=32, size=256, views=1, renders=1, label="configs[4] shape, one view per GPU"),
    "ref16k": dict(P=16384, F=3, size=128, views=1, renders=2,
                   label="ManiGaussian's own step (neural_rendering.py:386-393): 16 384 Gaussians, 2 renders"),
    "c4": dict(P=100000, F=32, size=128, views=4, timesteps=4, renders=1, deform=True,
               label="configs[3]: deformation MLP (fp32), 4 timesteps x 4 views = 16 renders per step shared by the ranks"),
    "c5": dict(P=500000, F=32, size=256, views=8, timesteps=1, renders=1, deform=True,
               label="configs[4]: 500k Gaussians + deformation MLP (fp32), 8 views per step shared by the ranks"),
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--config", default="c3", choices=sorted(CONFIGS))
    ap.add_argument("--mode", default="graph", choices=["graph", "eager", "eager-st"])
    ap.add_argument("--timesteps", type=int, default=None, help="dynamic configs: timesteps per step in total (c4: 4)")
    ap.add_argument("--P", type=int, default=None)
    ap.add_argument("--F", type=int, default=None)
    ap.add_argument("--size", type=int, default=None)
    ap.add_argument("--views", type=int, default=None, help="views of the Gaussian set each GPU renders per step (> 1: one "
                    "batched call, SURVEY 8f row 1; the headline config is 1)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--only-mode", action="store_true", help="time --mode only (profiling runs)")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--tight-bins", type=int, default=None)
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only to exercise "
                                                      "the N>1 path on a box with fewer GPUs than ranks)")
    ap.add_argument("--allreduce", default="dense", choices=["dense", "sparse"],
                    help="static configs, N > 1: 'sparse' reduces only the rows of Gaussians some rank saw (radii > 0): "
                         "parallel.sparse_all_reduce_grads -- one host read of the row count per step")
    ap.add_argument("--one-device", action="store_true", help="testing only: every rank uses cuda:0")
    ap.add_argument("--calibrate", action="store_true", help="counter passes: launch the library's known-instruction-mix "
                                                             "kernel a few times first (scripts/sq_counters.py checks it)")
    return ap.parse_args()


def self_launch(args):
    """`python bench.py --gpus N` with N > 1 and no launcher: re-execute under torch.distributed.run, one rank per GPU."""
    import torch
    if not args.one_device and torch.cuda.device_count() < args.gpus:
        raise SystemExit(f"bench.py --gpus {args.gpus}: only {torch.cuda.device_count()} HIP device(s) visible")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    raise SystemExit(subprocess.call(cmd, env=env))


def cpu_baseline(syn, sc, cam, d_color, d_feat, P, max_seconds):
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


def lib_hash():
    """Identity of the kernels being timed: the hash of its sources the library carries (mgs_build_id, baked in at compile
    time by csrc/Makefile) -- of the BINARY that runs, not of the working tree."""
    from manigaussian_amd import _lib
    return _lib.build_id()


COUNTER_FILES = ("r03_sq_counters.json", "r02_sq_counters.json")


def committed_counters(kernel_substr, build_id):
    """Per-launch counters of a kernel from the committed rocprofv3 passes (profiles/r03_sq_counters.json, written by
    scripts/sq_counters.py from runs of THIS command) -- only if they were collected from the binary being timed."""
    why = "no committed counters"
    for name in COUNTER_FILES:
        try:
            with open(os.path.join(ROOT, "profiles", name)) as f:
                j = json.load(f)
        except (OSError, ValueError) as e:
            why = f"{name}: {e}"
            continue
        have = j.get("lib_build_id") or j.get("lib_sha256_16")
        if have != build_id:
            why = f"profiles/{name} was collected from library {have}, timing {build_id}"
            continue
        for k, v in j.get("kernels", {}).items():
            if kernel_substr in k:
                return v, None
        why = f"kernel not in profiles/{name}"
    return None, why


def main():
    args = parse()
    env_world = os.environ.get("WORLD_SIZE")
    if env_world is None and args.gpus > 1:
        self_launch(args)
    import torch
    import torch.distributed as dist
    from manigaussian_amd import GaussianRasterizationSettings, GaussianRasterizer, _lib, check_status
    from manigaussian_amd import synthetic as syn
    from manigaussian_amd.parallel import all_reduce_grads, flat_alias, sparse_all_reduce_grads

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(env_world or "1")
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (there is no CPU fallback for the product path)")
    if args.one_device:
        local_rank = 0
    elif torch.cuda.device_count() <= local_rank:
        raise SystemExit(f"rank {rank}: LOCAL_RANK {local_rank} but {torch.cuda.device_count()} device(s) visible")
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

    cfg = dict(CONFIGS[args.config])
    for k in ("P", "F", "size", "views", "timesteps"):
        if getattr(args, k) is not None:
            cfg[k] = getattr(args, k)
    P, F, W, H, V, NR = cfg["P"], cfg["F"], cfg["size"], cfg["size"], max(1, cfg["views"]), cfg["renders"]
    deform = bool(cfg.get("deform"))
    M = 4
    sc = syn.make_scene(P, F=F, M=M, seed=0)  # identical on every rank: the replicated Gaussian set
    params = {k: v.to(dev).requires_grad_(True) for k, v in sc.items()}
    my_items = []  # dynamic configs: this rank's share of the step's (timestep, view) items
    if deform:
        # STRONG scaling: the step's work -- T timesteps x V views -- is fixed; rank r takes a contiguous share of the items,
        # so that the views of one timestep stay together (one MLP evaluation + one batched render per timestep and rank)
        T_total, V_total = max(1, cfg.get("timesteps", 1)), V
        items = [(t, v) for t in range(T_total) for v in range(V_total)]
        if len(items) % n_gpus:
            raise SystemExit(f"bench.py --config {args.config}: {len(items)} (timestep, view) items do not divide over "
                             f"{n_gpus} ranks")
        per = len(items) // n_gpus
        my_items = items[rank * per:(rank + 1) * per]
        cams = syn.circle_cameras(max(V_total, 8), W, H, negative_focal=True)
        renders_total = len(items)
    else:
        cams = syn.circle_cameras(max(n_gpus * V * NR, 8), W, H, negative_focal=True)
        my_cams = [cams[(rank * V * NR + i) % len(cams)] for i in range(V * NR)]
        renders_total = V * NR * n_gpus
    cam = cams[0] if deform else my_cams[0]
    d_color_h, d_feat_h = syn.make_cotangents(W, H, F, seed=1 + rank)
    plist = list(params.values())
    if not deform:
        means2D = torch.zeros(P, 3, device=dev, requires_grad=True)
        all_settings = [GaussianRasterizationSettings(**syn.camera_settings_kwargs(c, 1, True, device=dev)) for c in my_cams]
        rasts = [GaussianRasterizer(s) for s in all_settings]
        cots = [tuple(t.to(dev) for t in syn.make_cotangents(W, H, F, seed=1 + rank * V * NR + i)) for i in range(V * NR)]
        if V > 1:  # one batched call per step: V views of the same Gaussian set, gradients summed over the views on the device
            from manigaussian_amd import GaussianRasterizerBatch
            rast_batch = GaussianRasterizerBatch(all_settings[:V])
            d_color = torch.stack([c for c, _ in cots[:V]])
            d_feat = torch.stack([f for _, f in cots[:V]])
    else:
        from manigaussian_amd import GaussianRasterizerBatch
        from manigaussian_amd.deform import DeformationField, tune_gemms
        from manigaussian_amd.parallel import GradBucket
        if not os.environ.get("MGS_NO_GEMM_TUNING"):
            tune_gemms()  # TunableOp: the warm-up steps time hipBLASLt / rocBLAS candidates per GEMM shape (fp32 either way)
        g = torch.Generator().manual_seed(3)
        point_latent = torch.randn(P, 128, generator=g).to(dev).requires_grad_(True)
        z_feature = torch.randn(P, 39, generator=g).to(dev)
        field = DeformationField().to(dev)
        with torch.no_grad():  # the reference zero-initialises fc_1; give the deltas some life without exploding the scene
            for p_ in field.parameters():
                p_.mul_(0.05)
        # the MLP's parameter gradients live in ONE flat buffer (the only thing a trainer reduces: point_latent is an
        # activation of the voxel encoder, a local leaf here); autograd accumulates into it in place over the timesteps
        # (two of them alternate when N > 1, so that one step's asynchronous all-reduce may overlap the next step)
        buckets = [GradBucket(dict(field.named_parameters())) for _ in range(2 if world > 1 else 1)]
        bucket_turn = [0]
        plist = list(field.parameters())
        all_settings = [GaussianRasterizationSettings(**syn.camera_settings_kwargs(c, 1, True, device=dev)) for c in cams]
        groups = []  # (timestep, batched rasterizer over this rank's views of it, action, targets)
        for t in sorted({t for t, _ in my_items}):
            vs = [v for tt, v in my_items if tt == t]
            gt = torch.Generator().manual_seed(100 + t)
            groups.append(dict(t=t, views=vs, rast=GaussianRasterizerBatch([all_settings[v] for v in vs]),
                               action=torch.randn(1, 8, generator=gt).to(dev),
                               tgt_c=torch.rand(V_total, 3, H, W, generator=gt)[vs].to(dev),
                               tgt_f=torch.randn(V_total, F, H, W, generator=gt)[vs].to(dev)))
        n_c, n_f = float(renders_total * 3 * H * W), float(renders_total * F * H * W)  # the loss is a mean over ALL renders

    last_radii = [None]

    def render_once(i):
        if V > 1:
            return rast_batch(params["means3D"], None, params["opacities"], shs=params["shs"],
                              language_feature_precomp=params["language_feature"], scales=params["scales"],
                              rotations=params["rotations"])
        return rasts[i](means3D=params["means3D"], means2D=means2D, opacities=params["opacities"], shs=params["shs"],
                        language_feature_precomp=params["language_feature"], scales=params["scales"],
                        rotations=params["rotations"])

    def compute_step():
        """forward + backward of this rank's render(s); returns the gradients (views of ONE allocation per render)."""
        if deform:
            bucket = buckets[bucket_turn[0] % len(buckets)]
            bucket_turn[0] += 1
            bucket.attach()       # zero the flat MLP-gradient buffer, point every .grad at its view
            point_latent.grad = None
            for gr in groups:     # one timestep: MLP -> apply -> one batched render of this rank's views -> backward
                nxt = field(point_latent, z_feature, params["means3D"].detach(), params["shs"].detach(),
                            params["rotations"].detach(), params["scales"].detach(), params["opacities"].detach(),
                            action=gr["action"])
                color, feat, _ = gr["rast"](nxt["xyz"], None, nxt["opacity"], shs=nxt["sh"],
                                            language_feature_precomp=params["language_feature"].detach(),
                                            scales=nxt["scale"], rotations=nxt["rot"])
                loss = ((color - gr["tgt_c"]) ** 2).sum() / n_c + 0.01 * ((feat - gr["tgt_f"]) ** 2).sum() / n_f
                loss.backward()
            return [bucket.flat]
        out = None
        for i in range(NR):
            color, feat, radii = render_once(i)
            last_radii[0] = radii
            dc, df = (d_color, d_feat) if V > 1 else cots[i]
            gs = torch.autograd.grad([color, feat], plist, [dc, df])
            out = gs if out is None else out  # NR > 1: the renders are independent; the last ones' gradients stand in
        return out

    # ---- one "stepper" per mode: step() enqueues a whole step, grads() are the tensors an all-reduce takes ----
    class Eager:
        def __init__(self):
            self.last = None

        def next_slot(self):
            return None

        def step(self):
            self.last = compute_step()
            return self.last

    class Graphed:
        """The step captured into HIP graphs.  Two graphs with their own output buffers alternate when N > 1 so that one
        step's all-reduce may overlap the next step's replay."""

        def __init__(self, nbuf):
            for _ in range(3):  # learn the workspace marks, warm the allocator
                compute_step()
                check_status(dev)
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                compute_step()
            torch.cuda.current_stream().wait_stream(side)
            self.graphs, self.outs, self.i = [], [], 0
            for _ in range(nbuf):
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    o = compute_step()
                self.graphs.append(g)
                self.outs.append(o)

        def next_slot(self):
            return self.i % len(self.graphs)

        def step(self):
            k = self.i % len(self.graphs)
            self.i += 1
            self.graphs[k].replay()
            self.last = self.outs[k]
            return self.last

    pending = {}  # gradient buffer -> work handle of its all-reduce still in flight

    def run(stepper, k, collective=True):
        coll = world > 1 and collective
        for _ in range(k):
            slot = stepper.next_slot() if coll else None
            if coll and slot is not None:
                h = pending.pop(slot, None)
                if h is not None:
                    h.wait()  # the buffer this replay overwrites must have been reduced
            grads = stepper.step()
            if coll:
                # ONE in-place all-reduce of the allocation the parameter gradients alias, asynchronous on RCCL's stream:
                # it overlaps the next step (which writes another allocation)
                if slot is None:  # eager: every step has a fresh allocation; at most one all-reduce in flight
                    for h in pending.values():
                        if h is not None:
                            h.wait()
                    pending.clear()
                    slot = "eager"
                if args.allreduce == "sparse" and not deform:
                    vis = last_radii[0] > 0
                    sparse_all_reduce_grads(grads, vis.any(0) if vis.dim() == 2 else vis)
                    pending[slot] = None
                else:
                    pending[slot] = all_reduce_grads(grads, async_op=True)

    def sync_all():
        for h in list(pending.values()):
            if h is not None:
                h.wait()
        pending.clear()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    def timed(stepper, k, collective=True):
        """EXACTLY k steps between two (barrier + synchronize) brackets; max over ranks."""
        sync_all()
        t0 = time.perf_counter()
        run(stepper, k, collective)
        # completion is observed by polling an event before the closing (barrier + synchronize) bracket: a blocking
        # synchronize sleeps and wakes tens of microseconds after the GPU is done, which a 20-step region of 3 ms feels
        done = torch.cuda.Event()
        done.record()
        while not done.query():
            pass
        sync_all()
        el = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([el], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        return el

    extra_warmup = {}

    def measure(mode, steps, warmup, collective=True):
        torch.autograd.set_multithreading_enabled(mode != "eager-st")
        stepper = Graphed(2 if world > 1 else 1) if mode == "graph" else Eager()
        run(stepper, warmup, collective)
        # the W warm-up steps of the driver's short form are 0.8 ms of GPU work after seconds of host-only set-up: the device
        # is still ramping its clocks when the timed region starts.  More of the same steps, untimed, until ~30 ms have passed
        # (reported as `extra_warmup_steps`; the K timed steps and the W of the contract are untouched)
        est = timed(stepper, 2, collective) / 2  # (max over ranks: every rank derives the same count -- the steps hold collectives)
        extra = 2 + max(0, min(300, int(EXTRA_WARMUP_MS * 1e-3 / max(est, 1e-6))) - 2)
        run(stepper, extra - 2, collective)
        extra_warmup[mode] = extra
        el = timed(stepper, steps, collective)  # EXACTLY the K steps that were asked for
        long_run = None
        if el * 1e3 < MIN_TIMED_MS:  # a short region carries the brackets' overhead: measure a longer one BESIDE it
            k2 = int(steps * MIN_TIMED_MS / (el * 1e3) * 1.2) + 1
            long_run = (timed(stepper, k2, collective), k2)
        check_status(dev)
        torch.autograd.set_multithreading_enabled(True)
        return el, steps, stepper, long_run

    if args.calibrate:
        sink = torch.empty(256 * 1024, device=dev)
        for _ in range(3):
            _lib.check(_lib.lib().mgs_calibration_kernel(1000, sink.data_ptr(), None), "calibration")
        torch.cuda.synchronize()
    modes = [args.mode] if args.only_mode else [args.mode] + [m for m in ("graph", "eager", "eager-st") if m != args.mode]
    results, errors, long_runs = {}, {}, {}
    headline_stepper = None
    for m in modes:
        try:
            el, k, stp, lr = measure(m, args.steps, args.warmup)
            results[m] = (el, k)
            long_runs[m] = lr
            if m == args.mode:
                headline_stepper = stp
        except Exception as e:  # e.g. a graph capture the runtime refuses: report it, keep the other modes
            if m == args.mode and m != "graph":
                raise
            errors[m] = f"{type(e).__name__}: {e}"[:300]
    # the dominant kernel timed live: hipEvents around the render backward on its launch stream (eager steps: events
    # cannot be read back from inside a captured graph)
    torch.autograd.set_multithreading_enabled(False)
    _lib.profile_read(reset=True)
    _lib.set_option("profile", 1)
    e1 = Eager()
    run(e1, min(max(args.steps, 20), 200), collective=False)
    sync_all()
    _lib.set_option("profile", 0)
    prof = _lib.profile_read(reset=True)
    torch.autograd.set_multithreading_enabled(True)
    mode = args.mode if args.mode in results else next(iter(results))
    elapsed, steps = results[mode]

    exposed_ms, ar_bytes = None, 0
    if world > 1:
        fa = flat_alias(headline_stepper.last) if headline_stepper is not None else None
        ar_bytes = int(fa.numel() * 4) if fa is not None else int(sum(g.numel() for g in headline_stepper.last) * 4)
    if world > 1:  # what the collective costs on top of the compute: the same K steps without it
        el0, k0, _, lr0 = measure(mode, steps, 5, collective=False)
        lr1 = long_runs.get(mode)
        with_ms = (lr1[0] / lr1[1]) if lr1 else elapsed / steps
        without_ms = (lr0[0] / lr0[1]) if lr0 else el0 / k0
        exposed_ms = max(0.0, with_ms - without_ms) * 1e3

    # stage breakdown, untimed extra pass (eager: the stage timers are host-side event records)
    _lib.set_option("profile", 2)
    e = Eager()
    run(e, min(steps, 20), collective=False)
    sync_all()
    _lib.set_option("profile", 0)
    stages = {k: (ms / max(c, 1)) for k, (ms, c) in _lib.profile_read(reset=True).items()}

    # measured instance count (R) of what ONE launch of the render kernels covers on this rank
    from manigaussian_amd import _C
    launches_per_step = NR
    with torch.no_grad():
        et = torch.empty(0, device=dev)

        def count(xyz, rot, st):
            return _C.rasterize_gaussians(st.bg, xyz, et, params["language_feature"], params["opacities"], params["scales"],
                                          rot, 1.0, et, st.viewmatrix, st.projmatrix, st.tanfovx, st.tanfovy, H, W,
                                          params["shs"], 1, st.campos, False, False, True)[0]
        if deform:  # the first timestep's batch of views stands for a launch
            gr = groups[0]
            nxt = field(point_latent, z_feature, params["means3D"], params["shs"], params["rotations"], params["scales"],
                        params["opacities"], action=gr["action"])
            R = sum(count(nxt["xyz"], nxt["rot"], all_settings[v]) for v in gr["views"])
            launch_views, launches_per_step = len(gr["views"]), len(groups)
        else:
            R = sum(count(params["means3D"], params["rotations"], st) for st in all_settings[:V])
            launch_views = V
    dev_ids = [None] * world
    if world > 1:
        dist.all_gather_object(dev_ids, torch.cuda.current_device())
    else:
        dev_ids = [torch.cuda.current_device()]

    if rank == 0:
        ms_step = elapsed / steps * 1e3
        renders = renders_total // n_gpus            # renders per GPU per step
        value = P * renders_total * steps / elapsed  # whole job: every render of every rank
        bwd_ms, bwd_n = prof["render_bwd"]
        bwd_avg_ms = bwd_ms / max(bwd_n, 1)
        npix = W * H * launch_views  # pixels one launch covers
        bytes_k8 = R * (112 + 12 * F) + npix * (20 + 4 * F)
        achieved = bytes_k8 / (bwd_avg_ms * 1e-3) / 1e9 if bwd_avg_ms > 0 else 0.0
        bytes_path = (launch_views * P * (434 + 48 * M + 4 * F) + R * (196 + 16 * F) + npix * (40 + 8 * F)) * launches_per_step
        so_hash = lib_hash()
        def roof_block(label, substr, avg_ms, launches, bytes_alg):
            """What bounds a kernel, from the committed counter passes of THIS binary (profiles/r03_sq_counters.json): no
            throughput roof is near -- the waves spend their cycles waiting on dependent instructions and on memory / LDS /
            barriers (SQ_WAIT_*).  The HBM line (algorithmic and counter bytes) is kept beside the issue numbers."""
            cnt, why = committed_counters(substr, so_hash)
            traffic = cnt.get("hbm_bytes_per_launch") if cnt else None
            ach = bytes_alg / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
            cycles = avg_ms * 1e-3 * 2.4e9  # upper bound: 2.4 GHz peak clock
            # the contract's block: the HBM line of this kernel -- ALGORITHMIC bytes per launch over the launch
            # duration measured live with hipEvents on the launch stream, against 8 TB/s; `traffic` = HBM bytes per launch by
            # the counters (2 * FETCH_SIZE + WRITE_SIZE KiB, the guide's gfx950 correction) from the committed passes
            rb = {"kernel": label, "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                  "frac": ach / HBM_PEAK_GBS, "traffic": traffic, "avg_launch_ms": avg_ms, "launches": launches,
                  "algorithmic_bytes_per_launch": bytes_alg,
                  "frac_by_counter_bytes": (traffic / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if traffic and avg_ms > 0 else None,
                  "counters_note": why, "lib_build_id": so_hash}
            if cnt and cnt.get("SQ_INSTS_VALU") and cycles > 0:
                # beside it, what the counters say actually limits the kernel: no throughput roof is near, the waves wait
                gips = cnt["SQ_INSTS_VALU"] / (avg_ms * 1e-3) / 1e9
                wc = cnt.get("SQ_WAVE_CYCLES") or 0.0
                rb["limiter"] = {
                    "summary": "latency (dependent issue + waits): HBM, VALU issue and the matrix pipe are all far from peak",
                    "valu_issue": {"achieved": gips, "peak": VALU_PEAK_GIPS, "unit": "G wave-instr/s", "frac": gips / VALU_PEAK_GIPS},
                    "valu_busy_frac_of_simd_cycles": (cnt.get("SQ_ACTIVE_INST_VALU", 0.0) * 4.0) / (1024 * cycles),
                    "mfma_busy_frac_of_simd_cycles": cnt.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (1024 * cycles),
                    "wave_cycles_split": {k: (cnt[k] / wc if wc and k in cnt else None)
                                          for k in ("SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY")},
                    "per_launch": {k: cnt.get(k) for k in ("SQ_WAVES", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_MFMA",
                                                           "SQ_INSTS_LDS", "SQ_LDS_BANK_CONFLICT", "SQ_INSTS_VMEM_RD",
                                                           "SQ_INSTS_VMEM_WR")}}
            return rb

        roof = roof_block("K8 render backward (gm_bwd_kernel)", "gm_bwd_kernel", bwd_avg_ms, bwd_n, bytes_k8)
        bytes_k7 = R * (40 + 4 * F) + npix * (4 * (3 + F) + 8)
        roof_fwd = roof_block("K7 render forward (coop_fwd_pairs_kernel)", "coop_fwd_pairs_kernel",
                              stages.get("render_fwd", 0.0), min(steps, 20), bytes_k7)
        out = {
            "metric": "Gaussians rasterized/sec (fwd+bwd), 128x128, 32 feat-ch; HBM GB/s vs peak",
            "value": value, "unit": "Gaussians/s", "n_gpus": n_gpus, "steps": steps, "warmup": args.warmup,
            "extra_warmup_steps": extra_warmup.get(mode, 0),
            "ms_per_step": ms_step, "higher_is_better": True, "scaling": "strong" if deform else "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic", "mode": mode,
            "long_run": ({"steps": long_runs[mode][1], "ms_per_step": long_runs[mode][0] / long_runs[mode][1] * 1e3,
                          "value": P * renders_total * long_runs[mode][1] / long_runs[mode][0],
                          "why": f"the {steps}-step region lasted under {MIN_TIMED_MS:.0f} ms; the same stepper over a longer "
                                 "region (the short one carries ~0.1 ms of barrier + synchronize overhead)"}
                         if long_runs.get(mode) else None),
            "config": {"workload": f"{cfg['label']}: {P} Gaussians, {W}x{H}, RGB via SH deg 1 (M=4) + {F}-ch language "
                                   f"feature, fwd+bwd, {renders} render{'s' if renders > 1 else ''} per GPU per step"
                                   f"{' (batched calls)' if (V > 1 or deform) else ''}, negative-focal look-at cameras",
                       "name": args.config, "P": P, "W": W, "H": H, "F": F, "M": M,
                       "views_per_gpu": (renders if deform else V), "timesteps_per_step": cfg.get("timesteps", 1) if deform else 1,
                       "renders_per_step_per_gpu": renders, "renders_per_step_total": renders_total,
                       "render_launch_views": launch_views, "num_rendered_R": int(R), "R_over_P": R / (P * launch_views),
                       "tight_bins": _lib.get_option("tight_bins"),
                       "deformation": ("DeformationField per timestep: HIP input assembly -> fp32 MLP 70->512x5->7 (torch GEMMs, "
                                       "fused HIP elementwise passes) -> HIP apply; gradients of the MLP parameters (one flat "
                                       "bucket) and of point_latent") if deform else None,
                       "strong_scaling_recipe": (f"python bench.py --config {args.config} --gpus N for N in "
                                                 f"{[n for n in (1, 2, 4, 8, 16) if renders_total % n == 0]}: the step's "
                                                 f"{renders_total} renders are shared, {renders_total}/N per GPU") if deform else None,
                       "collective": ("none" if n_gpus == 1 else
                                      "1 asynchronous all-reduce of the flat MLP-gradient bucket per step (point_latent is a "
                                      "local leaf)" if deform else
                                      "1 in-place all-reduce of the per-Gaussian parameter gradients per step, overlapping the "
                                      "next step")},
            "modes_ms_per_step": {m: el / k * 1e3 for m, (el, k) in results.items()}, "mode_errors": errors or None,
            "distributed": {"backend": args.backend if world > 1 else None,
                            "world_size_seen": dist.get_world_size() if world > 1 else 1, "device_ids": dev_ids,
                            "allreduce_bytes_per_step": ar_bytes,
                            "allreduce_exposed_ms_per_step": exposed_ms},
            "roofline": roof, "roofline_fwd": roof_fwd,
            "path_hbm": {"algorithmic_bytes_per_step_per_gpu": bytes_path,
                         "achieved_GBps": bytes_path * n_gpus / (ms_step * 1e-3) / 1e9,
                         "frac_of_peak": bytes_path / (ms_step * 1e-3) / 1e9 / HBM_PEAK_GBS,
                         "note": "rasterizer path only (SURVEY.md 8d); the deformation MLP's GEMMs are MFMA work, not in it"
                                 if deform else "SURVEY.md 8d, whole rasterizer path"},
            "stages_ms": stages,
        }
        if not args.no_cpu_baseline and n_gpus == 1 and not deform:
            out["cpu_baseline"] = cpu_baseline(syn, sc, cam, d_color_h, d_feat_h, P, args.cpu_seconds)
            out["gpu_over_cpu"] = value / out["cpu_baseline"]["value"]
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
