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

Modes (all are timed and reported under "modes_ms_per_step"; `value` comes from --mode, default graph):
  graph     the step captured once with torch.cuda.graph through the PUBLIC autograd API and replayed (possible because
            nothing in the library synchronises; needs --forward-mode async, the opt-in).  The headline: a replay needs ~10 us
            of host time per step, so a 20-step region of 3 ms does not feel a host hiccup; one replay costs ~5 us more than
            the kernels.
  eager-safe  K plain Python steps through the public autograd API under the PACKAGE DEFAULTS -- forward mode "safe" (this
            shape waits for the preprocess's report: nothing speculative), torch's default autograd threading, the compiled
            binding (csrc/mgs_torch.cpp): what an unmodified caller of the drop-in gets, no opt-in of any kind.  Since the
            binding cut the host cost of a step to ~80 us this is GPU-bound at every BASELINE shape and usually a few us FASTER
            than the graph (0.153-0.156 vs 0.155-0.160 ms at configs[2]); it is not the headline only because a host that waits
            for the device once per step has no queue to absorb a hiccup (one 20-step run in ten read 0.18).
  eager-st  the Python step called K times with torch.autograd.set_multithreading_enabled(False): the backward is
            enqueued by the calling thread, the host runs ahead and the step is GPU-bound: ms_per_step == the sum of the
            kernel durations of rocprofv3 (profiles/).  A process-global switch, hence not the headline.
  eager     the same with torch's default autograd threading: every backward is handed to the engine's device thread and
            back (two thread wake-ups per step): host-bound at this step size, printed beside the headline.

EXACTLY --steps steps are timed for `value` / `ms_per_step` / `steps`.  A K-step region shorter than 50 ms is additionally
re-measured over a longer region and reported as `long_run` (the short region carries ~0.1 ms of bracket overhead).

Dynamic configs (BASELINE configs[3] / [4], strong scaling: the work of a step is fixed, the ranks share it --
manigaussian_amd.parallel.DynamicPlan):
  c4  100 000 Gaussians, DeformationField (fp32 MLP 70 -> 512 x 5 -> 7) per timestep, 4 timesteps x 4 views = 16 renders per
      step.  N <= 4 ranks: rank r evaluates timesteps r, r + N, ... alone; N = 8: two ranks share a timestep.
  c5  500 000 Gaussians, 256 x 256, F = 32, DeformationField, ONE timestep x 8 views per step: all N ranks share the timestep.
  Ranks that share a timestep split its MLP BY POINT (parallel.sharded_deformation: all-gather of the [P, 7] deltas,
  reduce-scatter of dL/d delta) and its views round-robin: the 500 000-point MLP (72 ms) is evaluated once per step across
  the job, not once per rank.  Per timestep on a rank: input assembly (HIP) of ITS points -> MLP GEMMs (torch / hipBLASLt)
  + fused elementwise passes (HIP) -> all-gather -> apply (HIP) -> ONE batched render of that rank's views -> l2(rgb) + 0.01
  l2(feature) -> backward: reduce-scatter, MLP backward into the flat parameter-gradient bucket (the one all-reduce,
  asynchronous) and into this rank's rows of point_latent (a local leaf, complete without a collective).

The JSON line also carries
  roofline      the dominant kernel (render backward) timed live with HIP events on its launch stream, against the HBM
                roof: `achieved` = the bytes THIS dataflow has to move per launch (every array once) / duration,
                `traffic` = counter bytes per launch from the committed passes of the same binary, `frac` <= 1 by
                construction (checked; --strict-roofline makes a violation fatal); beside it what actually limits the
                kernel (VALU issue / waits) and, for reference, SURVEY.md 8d's per-instance-atomic charge (not a bound);
  roofline_by_kernel  the same HBM line for all seven kernels of a step, counter-based fractions included;
  roofline_mlp  (dynamic configs) the deformation MLP against the fp32 MFMA roof (157.3 TFLOP/s): FLOPs and time of its
                forward + backward, measured apart from the rasterizer;
  cpu_baseline  Oracle B (oracle/mgs_oracle.c, a port: the reference has no CPU rasterizer) on the host cores, same
                workload, a bounded number of fwd+bwd passes; dynamic configs: + the MLP in torch on the same cores over a
                bounded sample of points.
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
MFMA_F32_PEAK_TFLOPS = 157.3  # same guide: dense fp32 on the matrix cores (v_mfma_f32_32x32x2_f32: 256 FLOP/clk/CU)
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
    ap.add_argument("--mode", default="graph", choices=["graph", "eager", "eager-st", "eager-safe", "eager-ctypes"])
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
    ap.add_argument("--fast-exp", type=int, default=None, help="MgsOptions.fast_exp of every call (default: the library's)")
    ap.add_argument("--bin-mode", type=int, default=None, help="MgsOptions.bin_mode of every call (0: binning tables in memory)")
    ap.add_argument("--dbg", type=int, default=None, help="MgsOptions.dbg of every call (diagnostic A/B switches, e.g. 32768 = the bin "
                                                          "scatter launch writes the keys instead of the preprocess)")
    ap.add_argument("--gm-waves", type=int, default=None, help="MgsOptions.gm_waves of every call (render backward: 12 = "
                                                               "two pixels per step, the default; 16 / 8 = the one-pixel forms)")
    ap.add_argument("--forward-mode", default="async", choices=["async", "safe", "blocking"],
                    help="manigaussian_amd.set_forward_mode: the bench opts into 'async' (speculative workspace sizing, no "
                         "host-device synchronisation: what graph capture needs); 'safe' is the package default")
    ap.add_argument("--strict-roofline", action="store_true", help="exit non-zero if a printed roofline fraction exceeds 1")
    ap.add_argument("--dry-collectives", action="store_true",
                    help="N = 1 only: create an RCCL process group of ONE rank and issue the exact collective call sequence of "
                         "the N > 1 path (sub-group creation, padded all-gather / reduce-scatter, alternating-bucket async "
                         "all-reduce) on the device: the RCCL branches run on hardware; what is measured is the calls' cost, "
                         "not transport")
    ap.add_argument("--no-reference-kernels", action="store_true", help="skip timing oracle/_ref (the reference's own kernels) "
                                                                        "beside the cpu_baseline")
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


def cpu_baseline(syn, sc, cam, d_color, d_feat, P, max_seconds, threads=None, max_passes=5):
    """Oracle B fwd+bwd on the host cores (the checker, timed beside the GPU path; never the product).  threads: OpenMP threads
    (None: all the host has)."""
    from oracle import oracle_b
    oracle_b.build()
    st = types.SimpleNamespace(**syn.camera_settings_kwargs(cam, 1, True))
    all_cores = oracle_b.max_threads()
    cores = all_cores if threads is None else min(int(threads), all_cores)
    oracle_b.set_threads(cores)
    try:
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
            if len(times) >= max_passes or time.perf_counter() - t_start > max_seconds:
                break
    finally:
        oracle_b.set_threads(all_cores)
    best = min(times)
    return {"value": P / best, "unit": "Gaussians/s", "cores": cores, "kind": "port",
            "sample": f"{len(times)} fwd+bwd passes of the same workload (1 view), best of {len(times)}: "
                      f"{best * 1e3:.1f} ms, OpenMP {cores} threads"}


def reference_kernels(syn, sc, cam, d_color, d_feat, P, F, value):
    """The REFERENCE's own kernels (oracle/_ref: forward.cu / backward.cu / rasterizer_impl.cu compiled with hipcc for gfx950,
    rebuilt at this feature width) timed on THIS GPU on the same workload, inputs resident, no PyTorch on their side
    (oracle/ref_wrapper.cu ref_bench) -- the meaningful same-node comparison beside the CPU port.  Like cpu_baseline this is the
    checker being timed, never the product; None where the library was not built (it is prebuilt in the development container
    from /root/reference and travels with the snapshot)."""
    try:
        from oracle import ref_cuda
        if not ref_cuda.available(F):
            return {"value": None, "note": f"oracle/_ref has no build for {F} feature channels here"}
        st = types.SimpleNamespace(**syn.camera_settings_kwargs(cam, 1, True))
        r = ref_cuda.bench(sc["means3D"], sc["opacities"], st, d_color, d_feat, warmup=3, iters=20, shs=sc["shs"],
                           language_feature=sc["language_feature"], scales=sc["scales"], rotations=sc["rotations"])
        v = P / r["ms_step"] * 1e3
        return {"ms_per_step": r["ms_step"], "ms_fwd": r["ms_fwd"], "ms_bwd": r["ms_bwd"], "value": v, "unit": "Gaussians/s",
                "num_rendered": r["num_rendered"], "this_library_over_reference_kernels": value / v,
                "what": "the reference's CUDA kernels, unmodified, built by hipcc for gfx950 (oracle/Makefile), 20 fwd+bwd "
                        "iterations on this GPU, same scene / camera / cotangents"}
    except Exception as e:  # a baseline must never take the bench line down
        return {"value": None, "note": f"{type(e).__name__}: {e}"[:200]}


def cpu_baseline_dynamic(syn, sc, cam, d_color, d_feat, P, timesteps, renders_total, max_seconds):
    """Dynamic configs on the host cores: the deformation MLP in torch (same module, plain path, fwd + bwd) over a bounded
    SAMPLE of points, scaled linearly to P (the MLP is independent per point), plus Oracle B for one render of the full set;
    the step = timesteps x MLP + renders_total x render."""
    import torch
    from manigaussian_amd.deform import ResnetFC
    rast = cpu_baseline(syn, sc, cam, d_color, d_feat, P, max_seconds * 0.5)
    t_render = P / rast["value"]
    n = min(P, 20000)
    mlp = ResnetFC(70)
    x = torch.randn(n, 128 + 70, requires_grad=True)
    times, t_start = [], time.perf_counter()
    while True:
        t0 = time.perf_counter()
        delta, _ = mlp(x)
        delta.sum().backward()
        times.append(time.perf_counter() - t0)
        mlp.zero_grad(set_to_none=True)
        x.grad = None
        if len(times) >= 4 or time.perf_counter() - t_start > max_seconds * 0.5:
            break
    t_mlp = min(times) * (P / n)
    t_step = timesteps * t_mlp + renders_total * t_render
    return {"value": P * renders_total / t_step, "unit": "Gaussians/s", "cores": rast["cores"], "kind": "port",
            "sample": f"MLP (torch CPU, {torch.get_num_threads()} threads): {len(times)} fwd+bwd passes over {n} of {P} points, "
                      f"best {min(times) * 1e3:.0f} ms, scaled by P/{n} -> {t_mlp * 1e3:.0f} ms per timestep; rasterizer: "
                      f"{rast['sample']}; step = {timesteps} x MLP + {renders_total} x render = {t_step * 1e3:.0f} ms"}


def lib_hash():
    """Identity of the kernels being timed: the hash of its sources the library carries (mgs_build_id, baked in at compile
    time by csrc/Makefile) -- of the BINARY that runs, not of the working tree."""
    from manigaussian_amd import _lib
    return _lib.build_id()


def counter_files(cfg_name, views):
    """Committed counter passes for this workload (scripts/gpu_round4.sh writes them), newest round first."""
    suffix = "" if (cfg_name == "c3" and views == 1) else f"_{cfg_name}" + (f"_v{views}" if views > 1 else "")
    return [f"r{r:02d}_sq_counters{suffix}.json" for r in (6, 5, 4, 3, 2)]


def committed_counters(kernel_substr, build_id, files):
    """Per-launch counters of a kernel from the committed rocprofv3 passes (profiles/r04_sq_counters*.json, written by
    scripts/sq_counters.py from runs of THIS command) -- only if they were collected from the binary being timed."""
    why = "no committed counters"
    for name in files:
        try:
            with open(os.path.join(ROOT, "profiles", name)) as f:
                j = json.load(f)
        except (OSError, ValueError) as e:
            why = f"{name}: {type(e).__name__}"
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


# kernel (stage-timer name, substring of the kernel's symbol, label)
KERNELS_SEGSORT = [("preprocess_fwd", "preprocess_fwd_kernel", "K2 forward preprocess"),
                   ("bin_scatter", "bin_scatter_kernel", "K4' bin scatter"),
                   ("bin_segsort", "bin_segsort_kernel", "K5' segment sort"),
                   ("bin_merge", "bin_merge_emit_kernel", "K6' rank merge + emit"),
                   ("render_fwd", "coop_fwd_pairs_kernel", "K7 render forward"),
                   ("render_bwd", "gm_bwd_kernel", "K8 render backward"),
                   ("preprocess_bwd", "preprocess_bwd_kernel", "K9+K10 backward preprocess")]
# bin_mode 2 (the default since round 6): ONE bucket-rank launch in the segment sort's stage slot, no merge launch
KERNELS_BUCKET = [k for k in KERNELS_SEGSORT if k[0] not in ("bin_segsort", "bin_merge")]
KERNELS_BUCKET.insert(2, ("bin_segsort", "bin_bucket_emit_kernel", "K5'' bucket rank + emit (one launch; stage slot of the segment sort)"))


def model_bytes(P, V, M, F, npix, T, R, nvis, inc, pixel_chunks, nblk, bucket=True, direct=False):
    """HBM bytes ONE launch of each kernel has to move in THIS dataflow, every array once: P Gaussians, V views
    per launch (Pv = V P virtual Gaussians), nvis = sum over the views of Gaussians with radii > 0, R = (Gaussian, tile)
    instances, inc = (8x8 block, Gaussian) incidences of the chunks some pixel visited, pixel_chunks = (pixel, 64-survivor chunk)
    pairs visited (the unit of the forward -> backward state: 3 + F partial sums, T_end, T_mid, last_pos per pixel and
    chunk), npix pixels, T tiles, nblk = preprocess workgroups.  Nothing is charged per (pixel, Gaussian) pair or per instance for the gradient sums: they are
    reduced in registers / LDS and leave as one atomic row per (block, Gaussian), which the L2 merges -- one write-back per
    visible Gaussian.  These are lower bounds of what the kernel must move, so bytes / time <= the HBM roof."""
    Pv = V * P
    rec = 32 + 12 + 4 * F                      # packed record + rgb + feature row of one visible Gaussian
    state = pixel_chunks * 4 * (3 + F + 3)     # per (pixel, chunk): partial sums, T_end, T_mid, last_pos
    ncol = Pv if M else P                      # dL_dcolors rows: per (view, Gaussian) with SH colours
    zero = 4 * (8 * Pv + 3 * ncol + F * P)     # the backward's accumulator block, zeroed by the forward preprocess
    # direct binning (round 6): the preprocess writes the keys itself (and still its reservation rows, for a later
    # re-binning into another workspace); no scatter launch; the bucket rank reads the tile histogram
    return {
        "preprocess_fwd": Pv * 12 + nvis * (28 + 4 + 12 * M) + Pv * 16 + nvis * (4 + 32 + 12 + 24 + 1) + zero +
                          ((nblk * T * 4 + R * 8) if direct else 0),
        "bin_scatter": 0 if direct else Pv * 12 + nblk * T * 4 + R * 8,
        "bin_segsort": (R * 8 + R * 4 + (T * 4 if direct else 0)) if bucket else (R * 8 + R * 8),   # bucket rank: keys in, sorted ids out
        "bin_merge": 0 if bucket else (R * 8 + R * 4),
        "render_fwd": R * 4 + nvis * rec + inc * 4 + state + npix * (4 * (3 + F) + 8),
        "render_bwd": inc * 4 + nvis * rec + state + npix * (4 * (3 + F) + 8) + nvis * (32 + 12 + 4 * F),
        "preprocess_bwd": Pv * 4 + nvis * (12 + 28 + 12 * M + 24 + 1 + 32 + 12) + Pv * 12 + P * (4 + 12 + 24 + 12 * M + 12 + 16),
    }


def main():
    args = parse()
    env_world = os.environ.get("WORLD_SIZE")
    if env_world is None and args.gpus > 1:
        self_launch(args)
    import torch
    import torch.distributed as dist
    import manigaussian_amd
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
    dry = bool(args.dry_collectives) and world == 1
    if dry:  # an RCCL group of one rank: every collective of the N > 1 path is issued to the device (parallel._DRY)
        from manigaussian_amd import parallel as _par
        with socket.socket() as s_:
            s_.bind(("127.0.0.1", 0))
            port_ = s_.getsockname()[1]
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(port_))
        os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")
        dist.init_process_group(args.backend, rank=0, world_size=1, **({"device_id": dev} if args.backend == "nccl" else {}))
        _par.set_dry_collectives(True)
    coll_on = world > 1 or dry
    if args.tight_bins is not None:
        _lib.set_option("tight_bins", args.tight_bins)
    if args.fast_exp is not None:
        _lib.set_option("fast_exp", args.fast_exp)
    if args.bin_mode is not None:
        _lib.set_option("bin_mode", args.bin_mode)
    if args.dbg is not None:
        _lib.set_option("dbg", args.dbg)
    if args.gm_waves is not None:
        _lib.set_option("gm_waves", args.gm_waves)
    # The package default ("safe") never sizes a workspace speculatively; a training loop that wants a step without any
    # host-device synchronisation -- and HIP-graph capture -- opts into "async", as this benchmark does (--forward-mode).
    manigaussian_amd.set_forward_mode(args.forward_mode)

    cfg = dict(CONFIGS[args.config])
    for k in ("P", "F", "size", "views", "timesteps"):
        if getattr(args, k) is not None:
            cfg[k] = getattr(args, k)
    P, F, W, H, V, NR = cfg["P"], cfg["F"], cfg["size"], cfg["size"], max(1, cfg["views"]), cfg["renders"]
    deform = bool(cfg.get("deform"))
    M = 4
    sc = syn.make_scene(P, F=F, M=M, seed=0)  # identical on every rank: the replicated Gaussian set
    params = {k: v.to(dev).requires_grad_(True) for k, v in sc.items()}
    plan = None
    if deform:
        # STRONG scaling: the step's work -- T timesteps x V views -- is fixed and shared (parallel.DynamicPlan): ranks that
        # share a timestep split its MLP by point and its views round-robin
        from manigaussian_amd.parallel import DynamicPlan, GradBucket, sharded_deformation
        T_total, V_total = max(1, cfg.get("timesteps", 1)), V
        try:
            plan = DynamicPlan(T_total, V_total, rank, world)
        except ValueError as e:
            raise SystemExit(f"bench.py --config {args.config} --gpus {n_gpus}: {e}")
        cams = syn.circle_cameras(max(V_total, 8), W, H, negative_focal=True)
        renders_total = T_total * V_total
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
        if not os.environ.get("MGS_NO_GEMM_TUNING"):
            tune_gemms()  # TunableOp: the warm-up steps time hipBLASLt / rocBLAS candidates per GEMM shape (fp32 either way)
        g = torch.Generator().manual_seed(3)
        lo, hi = plan.point_rows(P)
        # point_latent is an activation of the voxel encoder, a LOCAL leaf: this rank holds (and gets the complete gradient
        # of) the rows of the points it evaluates -- all of them when it works alone on its timesteps
        point_latent = torch.randn(P, 128, generator=g)[lo:hi].to(dev).requires_grad_(True)
        z_feature = torch.randn(P, 39, generator=g)[lo:hi].to(dev)
        field = DeformationField().to(dev)
        with torch.no_grad():  # the reference zero-initialises fc_1; give the deltas some life without exploding the scene
            for p_ in field.parameters():
                p_.mul_(0.05)
        # the MLP's parameter gradients live in ONE flat buffer (the only thing a trainer all-reduces); autograd accumulates
        # into it in place over the timesteps (two alternate when N > 1: a step's asynchronous all-reduce may overlap the next)
        buckets = [GradBucket(dict(field.named_parameters())) for _ in range(2 if coll_on else 1)]
        bucket_turn = [0]
        plist = list(field.parameters())
        all_settings = [GaussianRasterizationSettings(**syn.camera_settings_kwargs(c, 1, True, device=dev)) for c in cams]
        groups = []  # (timestep, batched rasterizer over this rank's views of it, action, targets)
        for t in plan.timesteps:
            vs = plan.views
            gt = torch.Generator().manual_seed(100 + t)
            groups.append(dict(t=t, views=vs, rast=GaussianRasterizerBatch([all_settings[v] for v in vs]),
                               action=torch.randn(1, 8, generator=gt).to(dev),
                               tgt_c=torch.rand(V_total, 3, H, W, generator=gt)[vs].to(dev),
                               tgt_f=torch.randn(V_total, F, H, W, generator=gt)[vs].to(dev)))
        n_c, n_f = float(renders_total * 3 * H * W), float(renders_total * F * H * W)  # the loss is a mean over ALL renders
        if (plan.group_size > 1 or plan.group is not None) and args.mode == "graph":
            args.mode = "eager-st"  # collectives inside the step (all-gather / reduce-scatter): the step is enqueued eagerly

    last_radii = [None]

    def render_once(i):
        if V > 1:
            return rast_batch(params["means3D"], None, params["opacities"], shs=params["shs"],
                              language_feature_precomp=params["language_feature"], scales=params["scales"],
                              rotations=params["rotations"])
        return rasts[i](means3D=params["means3D"], means2D=means2D, opacities=params["opacities"], shs=params["shs"],
                        language_feature_precomp=params["language_feature"], scales=params["scales"],
                        rotations=params["rotations"])

    def deform_group(gr):
        """One timestep on this rank up to the rendered images: (point-sharded) MLP -> apply -> one batched render."""
        nxt = sharded_deformation(field, point_latent, z_feature, params["means3D"].detach(), params["shs"].detach(),
                                  params["rotations"].detach(), params["scales"].detach(), params["opacities"].detach(),
                                  action=gr["action"], group=plan.group)
        return gr["rast"](nxt["xyz"], None, nxt["opacity"], shs=nxt["sh"],
                          language_feature_precomp=params["language_feature"].detach(), scales=nxt["scale"],
                          rotations=nxt["rot"])

    def compute_step():
        """forward + backward of this rank's render(s); returns the gradients (views of ONE allocation per render)."""
        if deform:
            bucket = buckets[bucket_turn[0] % len(buckets)]
            bucket_turn[0] += 1
            bucket.attach()       # zero the flat MLP-gradient buffer, point every .grad at its view
            point_latent.grad = None
            for gr in groups:     # one timestep: MLP -> apply -> one batched render of this rank's views -> backward
                color, feat, radii = deform_group(gr)
                last_radii[0] = radii
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
        coll = coll_on and collective
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
        if coll_on:
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
            if args.backend == "gloo":
                t = t.cpu()
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        return el

    extra_warmup = {}

    from manigaussian_amd import _C as _shim

    def measure(mode, steps, warmup, collective=True):
        # eager-safe: the package's DEFAULT options (forward mode "safe", default autograd threading, compiled binding) --
        # what an unmodified caller gets; eager-ctypes: the round-4 host path (the ctypes shim) for comparison
        old_fm = manigaussian_amd.set_forward_mode("safe") if mode == "eager-safe" else None
        try:
            with _shim.use_compiled(mode != "eager-ctypes"):
                return _measure(mode, steps, warmup, collective)
        finally:
            if old_fm is not None:
                manigaussian_amd.set_forward_mode(old_fm)

    def _measure(mode, steps, warmup, collective=True):
        torch.autograd.set_multithreading_enabled(mode != "eager-st")
        stepper = Graphed(2 if coll_on else 1) if mode == "graph" else Eager()
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
    all_modes = ("graph", "eager", "eager-st", "eager-safe", "eager-ctypes")
    if deform and (plan.group_size > 1 or plan.group is not None):
        all_modes = ("eager", "eager-st", "eager-safe")
    modes = [args.mode] if args.only_mode else [args.mode] + [m for m in all_modes if m != args.mode]
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
            print(f"bench.py: mode {m} failed: {errors[m]}", file=sys.stderr)
    if not results:
        raise SystemExit(f"bench.py: no mode could be measured: {errors}")
    mode = args.mode if args.mode in results else next(iter(results))
    elapsed, steps = results[mode]

    # SURVEY.md 8d's protocol beside the bracketed mean: a hipEvent pair per step on the work stream, median over <= 100 steps
    median_ms = None
    if headline_stepper is not None:
        torch.autograd.set_multithreading_enabled(mode != "eager-st")
        n_med = min(max(steps, 20), 100)
        old_fm = manigaussian_amd.set_forward_mode("safe") if mode == "eager-safe" else None
        with _shim.use_compiled(mode != "eager-ctypes"):
            evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n_med)]
            sync_all()
            for e0, e1 in evs:
                e0.record()
                run(headline_stepper, 1, collective=True)
                e1.record()
            sync_all()
        if old_fm is not None:
            manigaussian_amd.set_forward_mode(old_fm)
        torch.autograd.set_multithreading_enabled(True)
        ts = sorted(e0.elapsed_time(e1) for e0, e1 in evs)
        median_ms = {"median": ts[len(ts) // 2], "min": ts[0], "p90": ts[int(len(ts) * 0.9)], "steps": n_med,
                     "what": "hipEvent pair around every step on the work stream (SURVEY.md 8d: median of <= 100); eager "
                             "modes: the span between the two records holds one step's kernels only when the host runs ahead"}
        check_status(dev)

    exposed_ms, ar_bytes = None, 0
    if coll_on:
        fa = flat_alias(headline_stepper.last) if headline_stepper is not None else None
        ar_bytes = int(fa.numel() * 4) if fa is not None else int(sum(g.numel() for g in headline_stepper.last) * 4)
    if coll_on:  # what the collective costs on top of the compute: the same K steps without it
        el0, k0, _, lr0 = measure(mode, steps, 5, collective=False)
        lr1 = long_runs.get(mode)
        with_ms = (lr1[0] / lr1[1]) if lr1 else elapsed / steps
        without_ms = (lr0[0] / lr0[1]) if lr0 else el0 / k0
        exposed_ms = max(0.0, with_ms - without_ms) * 1e3

    # every kernel timed live: hipEvents around each launch on its launch stream (eager steps: events cannot be read back
    # from inside a captured graph), an untimed extra pass
    torch.autograd.set_multithreading_enabled(False)
    _lib.profile_read(reset=True)
    _lib.set_option("profile", 2)
    e = Eager()
    n_prof = min(max(args.steps, 20), 100) if not deform else min(max(args.steps, 4), 10)
    run(e, n_prof, collective=False)
    sync_all()
    _lib.set_option("profile", 0)
    prof = _lib.profile_read(reset=True)
    torch.autograd.set_multithreading_enabled(True)
    stages = {k: (ms / max(c, 1)) for k, (ms, c) in prof.items()}

    # what ONE launch of the render kernels covers on this rank: instances (R), visible Gaussians, (block, Gaussian)
    # incidences and chunks of the forward -> backward state (mgs_forward_stats, a blocking diagnostic)
    import ctypes
    launches_per_step = len(groups) if deform else NR
    launch_views = len(groups[0]["views"]) if deform else V
    torch.cuda.synchronize()
    # (a forward of its own whose outputs stay alive: the autograd node keeps the workspaces the statistics are read from)
    with _shim.use_compiled(False):  # (the ctypes shim: its autograd node exposes the forward's handle)
        color_s, feat_s, radii_s = deform_group(groups[0]) if deform else render_once(0)
    torch.cuda.synchronize()
    handle = color_s.grad_fn.num_rendered  # the forward's ForwardHandle (manigaussian_amd/_C.py)
    R = handle.binned()    # (Gaussian, tile) instances the kernels actually move
    R_reference = int(handle)  # the reference's num_rendered (3-sigma rects), what the call hands back
    nvis = int((radii_s > 0).sum().item())
    inc_, ch_, pc_ = ctypes.c_int64(0), ctypes.c_int64(0), ctypes.c_int64(0)
    _lib.check(_lib.lib().mgs_forward_stats(ctypes.byref(handle.a), launch_views if (deform or V > 1) else 0,
                                            ctypes.byref(inc_), ctypes.byref(ch_), ctypes.byref(pc_), None), "mgs_forward_stats")
    incidences, chunks, pixel_chunks = int(inc_.value), int(ch_.value), int(pc_.value)
    del color_s, feat_s, radii_s, handle

    # the deformation MLP apart from the rasterizer: forward + backward of this rank's points, hipEvent-timed
    mlp_block = None
    if deform:
        from manigaussian_amd.deform import assemble_deform_input
        n_loc = int(point_latent.shape[0])
        gr = groups[0]
        lo, hi = plan.point_rows(P)

        def mlp_only():
            zx = assemble_deform_input(point_latent, z_feature, params["means3D"].detach()[lo:hi], params["shs"].detach()[lo:hi],
                                       params["rotations"].detach()[lo:hi], params["scales"].detach()[lo:hi],
                                       params["opacities"].detach()[lo:hi], None, gr["action"])
            delta, _ = field.mlp(zx)
            delta.backward(torch.ones_like(delta))
        for _ in range(2):
            mlp_only()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n_it = 5
        ev0.record()
        for _ in range(n_it):
            mlp_only()
        ev1.record()
        torch.cuda.synchronize()
        mlp_ms = ev0.elapsed_time(ev1) / n_it
        point_latent.grad = None
        macs = 70 * 512 + 3 * 128 * 512 + 10 * 512 * 512 + 512 * 7  # per point: lin_in, 3 lin_z, 5 blocks x 2, lin_out
        flops = 3 * 2 * macs * n_loc                                  # forward + data gradients + weight gradients
        mlp_block = {"bound": "mfma", "dtype": "f32", "achieved": flops / (mlp_ms * 1e-3) / 1e12,
                     "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                     "frac": flops / (mlp_ms * 1e-3) / 1e12 / MFMA_F32_PEAK_TFLOPS, "flops_per_timestep_per_rank": flops,
                     "ms_per_timestep_per_rank": mlp_ms, "points_per_rank": n_loc,
                     "what": "input assembly + ResnetFC forward + backward (fp32 GEMMs on hipBLASLt / rocBLAS through torch, "
                             "fused HIP elementwise passes) of this rank's points for one timestep, hipEvent-timed apart from "
                             "the rasterizer; FLOPs = 3 x 2 x 3.28 M MACs per point (the elementwise passes add time, no FLOPs)"}

    dev_ids = [None] * world
    if world > 1:
        dist.all_gather_object(dev_ids, torch.cuda.current_device())
    else:
        dev_ids = [torch.cuda.current_device()]

    if rank == 0:
        ms_step = elapsed / steps * 1e3
        renders = launches_per_step * launch_views   # renders per GPU per step
        value = P * renders_total * steps / elapsed  # whole job: every render of every rank
        npix = W * H * launch_views  # pixels one launch covers
        T_tiles = ((W + 15) // 16) * ((H + 15) // 16) * launch_views
        pre_block = 512 if P <= 131072 else 1024  # (csrc/mgs_common.h pre_block())
        nblk = launch_views * ((P + pre_block - 1) // pre_block)
        bucket = _lib.get_option("bin_mode") == 2 and T_tiles <= 4096
        direct = bucket and prof.get("bin_scatter", (0.0, 0))[1] == 0  # no scatter launch was timed: the preprocess wrote the keys
        KERNELS = [k for k in KERNELS_BUCKET if not (direct and k[0] == "bin_scatter")] if bucket else KERNELS_SEGSORT
        mb = model_bytes(P, launch_views, M, F, npix, T_tiles, R, nvis, incidences, pixel_chunks, nblk, bucket, direct)
        so_hash = lib_hash()
        cfiles = counter_files(args.config, launch_views if not deform else 1) if not deform else \
            [f"r06_sq_counters_{args.config}.json", f"r05_sq_counters_{args.config}.json", f"r04_sq_counters_{args.config}.json"]
        violations = []

        def hbm_line(stage, substr, label):
            """The HBM line of one kernel: model bytes per launch / its average launch duration (hipEvents on the launch
            stream) against 8 TB/s; `traffic` = HBM bytes per launch by the counters (2 * FETCH_SIZE + WRITE_SIZE KiB, the
            guide's gfx950 correction) from the committed passes of THIS binary, and the fraction they give."""
            avg_ms, (tot_ms, launches) = stages.get(stage, 0.0), prof.get(stage, (0.0, 0))
            cnt, why = committed_counters(substr, so_hash, cfiles)
            traffic = cnt.get("hbm_bytes_per_launch") if cnt else None
            ach = mb[stage] / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
            rb = {"kernel": label, "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                  "frac": ach / HBM_PEAK_GBS, "traffic": traffic, "avg_launch_ms": avg_ms, "launches": launches,
                  "algorithmic_bytes_per_launch": mb[stage],
                  "frac_by_counter_bytes": (traffic / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if traffic and avg_ms > 0 else None,
                  "traffic_over_algorithmic": (traffic / mb[stage]) if traffic and mb[stage] else None,
                  "counters_note": why}
            for k in ("frac", "frac_by_counter_bytes"):
                if rb[k] is not None and rb[k] > 1.0:
                    violations.append(f"{label}: {k} = {rb[k]:.3f}")
            return rb, cnt

        by_kernel = {}
        cnts = {}
        for stage, substr, label in KERNELS:
            by_kernel[stage], cnts[stage] = hbm_line(stage, substr, label)

        def limiter(stage):
            """What the counters say limits a render kernel: no throughput roof is near -- the waves spend their cycles
            waiting on dependent instructions and on memory / LDS / barriers (SQ_WAIT_*)."""
            cnt, avg_ms = cnts[stage], stages.get(stage, 0.0)
            if not (cnt and cnt.get("SQ_INSTS_VALU") and avg_ms > 0):
                return None
            cycles = avg_ms * 1e-3 * 2.4e9  # upper bound: 2.4 GHz peak clock
            gips = cnt["SQ_INSTS_VALU"] / (avg_ms * 1e-3) / 1e9
            wc = cnt.get("SQ_WAVE_CYCLES") or 0.0
            return {"summary": "latency (dependent issue + waits): HBM, VALU issue and the matrix pipe are all far from peak",
                    "valu_issue": {"achieved": gips, "peak": VALU_PEAK_GIPS, "unit": "G wave-instr/s", "frac": gips / VALU_PEAK_GIPS},
                    "valu_busy_frac_of_simd_cycles": (cnt.get("SQ_ACTIVE_INST_VALU", 0.0) * 4.0) / (1024 * cycles),
                    "mfma_busy_frac_of_simd_cycles": cnt.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (1024 * cycles),
                    "wave_cycles_split": {k: (cnt[k] / wc if wc and k in cnt else None)
                                          for k in ("SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY")},
                    "per_launch": {k: cnt.get(k) for k in ("SQ_WAVES", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_MFMA",
                                                           "SQ_INSTS_LDS", "SQ_LDS_BANK_CONFLICT", "SQ_INSTS_VMEM_RD",
                                                           "SQ_INSTS_VMEM_WR")}}

        roof = dict(by_kernel["render_bwd"], kernel="K8 render backward (gm_bwd_kernel)", lib_build_id=so_hash)
        roof["limiter"] = limiter("render_bwd")
        bwd_ms = stages.get("render_bwd", 0.0)
        s8d = R * (112 + 12 * F) + npix * (20 + 4 * F)
        roof["survey_8d_charge"] = {
            "bytes_per_launch": s8d, "frac_if_charged": (s8d / (bwd_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if bwd_ms > 0 else None,
            "note": "SURVEY.md 8d's K8 rows R(112+12F) + N(20+4F) charge the reference's per-INSTANCE read-modify-write "
                    "atomics, which this kernel never issues (register / LDS reduction, one atomic row per (block, Gaussian)): "
                    "not a bound of this dataflow -- it exceeds 1 at large R / P -- and kept only so that earlier rounds' "
                    "numbers stay comparable"}
        roof_fwd = dict(by_kernel["render_fwd"], kernel="K7 render forward (coop_fwd_pairs_kernel)")
        roof_fwd["limiter"] = limiter("render_fwd")
        path_model = sum(mb[k] for k, _, _ in KERNELS) * launches_per_step
        path_traffic = None
        if all(by_kernel[k]["traffic"] for k in by_kernel):
            path_traffic = sum(by_kernel[k]["traffic"] for k in by_kernel) * launches_per_step
        s8d_path = (launch_views * P * (434 + 48 * M + 4 * F) + R * (196 + 16 * F) + npix * (40 + 8 * F)) * launches_per_step
        rast_ms = sum(stages.get(k, 0.0) for k, _, _ in KERNELS) * launches_per_step  # the rasterizer's kernels per step
        path_frac = path_model / (rast_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if rast_ms > 0 else None
        if path_frac is not None and path_frac > 1.0:
            violations.append(f"path_hbm: frac = {path_frac:.3f}")
        if mlp_block is not None and mlp_block["frac"] > 1.0:
            violations.append(f"roofline_mlp: frac = {mlp_block['frac']:.3f}")
        out = {
            "metric": "Gaussians rasterized/sec (fwd+bwd), 128x128, 32 feat-ch; HBM GB/s vs peak",
            "value": value, "unit": "Gaussians/s", "n_gpus": n_gpus, "steps": steps, "warmup": args.warmup,
            "extra_warmup_steps": extra_warmup.get(mode, 0),
            "ms_per_step": ms_step, "higher_is_better": True, "scaling": "strong" if deform else "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic", "mode": mode, "forward_mode": manigaussian_amd.forward_mode(),
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
                       "num_rendered_reference": int(R_reference),
                       "visible_gaussians_per_launch": nvis, "block_gaussian_incidences_per_launch": incidences,
                       "chunks_per_launch": chunks, "pixel_chunks_per_launch": pixel_chunks,
                       "tight_bins": _lib.get_option("tight_bins"), "fast_exp": _lib.get_option("fast_exp"),
                       "bin_mode": _lib.get_option("bin_mode"), "direct_binning": bool(direct),
                       "deformation": ("DeformationField per timestep: HIP input assembly -> fp32 MLP 70->512x5->7 (torch GEMMs, "
                                       "fused HIP elementwise passes) -> HIP apply; gradients of the MLP parameters (one flat "
                                       "bucket) and of point_latent") if deform else None,
                       "partition": (dict(plan.describe(P), timesteps_total=plan.n_timesteps, views_total=plan.n_views,
                                          allreduce_bytes_per_step=ar_bytes) if deform else None),
                       "strong_scaling_recipe": (f"python bench.py --config {args.config} --gpus N: the step's "
                                                 f"{renders_total} renders are shared; ranks that share a timestep split its "
                                                 f"MLP by point (P/N points each at N > timesteps)") if deform else None,
                       "collective": ("none" if n_gpus == 1 else
                                      ("per timestep an all-gather of the [P,7] deltas and a reduce-scatter of dL/d delta "
                                       "inside the ranks that share it; " if plan.group_size > 1 else "") +
                                      "1 asynchronous all-reduce of the flat MLP-gradient bucket per step (point_latent is a "
                                      "local leaf)" if deform else
                                      "1 in-place all-reduce of the per-Gaussian parameter gradients per step, overlapping the "
                                      "next step")},
            "modes_ms_per_step": {m: el / k * 1e3 for m, (el, k) in results.items()}, "mode_errors": errors or None,
            "modes_note": "graph / eager / eager-st run with --forward-mode (default async: no host-device synchronisation, "
                          "what capture needs); eager-safe = the package DEFAULTS (forward mode safe: this shape waits for the "
                          "preprocess's report, default autograd threading): what an unmodified caller gets; eager-ctypes = "
                          "eager through the round-4 ctypes shim instead of the compiled binding",
            "host_binding": {"compiled": _shim.compiled() is not None,
                             "counters": _shim.compiled().counters() if _shim.compiled() is not None else None},
            "ms_per_step_events": median_ms,
            "distributed": {"backend": args.backend if coll_on else None,
                            "world_size_seen": dist.get_world_size() if coll_on else 1, "device_ids": dev_ids,
                            "dry_collectives": dry,
                            "allreduce_bytes_per_step": ar_bytes,
                            "allreduce_exposed_ms_per_step": exposed_ms,
                            "predicted_ring_allreduce_ms": ({str(n): 2 * (n - 1) / n * ar_bytes / 153e9 * 1e3 for n in (2, 4, 8)}
                                                            if ar_bytes else None),
                            "note": ("dry run on ONE device: the RCCL call sequence of the N > 1 path executed (process group "
                                     "of one rank); exposed time = API + kernel-launch cost of the collectives, no transport. "
                                     "predicted_ring_allreduce_ms = 2 (N-1)/N x bytes at 153 GB/s per xGMI link direction: a "
                                     "prediction, never measured by the builder (one GPU)") if dry else None},
            "roofline": roof, "roofline_fwd": roof_fwd, "roofline_by_kernel": by_kernel,
            "roofline_mlp": mlp_block,
            "path_hbm": {"algorithmic_bytes_per_step_per_gpu": path_model,
                         "kernel_ms_per_step_per_gpu": rast_ms,
                         "achieved_GBps": (path_model / (rast_ms * 1e-3) / 1e9) if rast_ms > 0 else None,
                         "frac_of_peak": path_frac,
                         "counter_bytes_per_step_per_gpu": path_traffic,
                         "frac_of_peak_by_counter_bytes": (path_traffic / (rast_ms * 1e-3) / 1e9 / HBM_PEAK_GBS)
                         if path_traffic and rast_ms > 0 else None,
                         "survey_8d_bytes_per_step_per_gpu": s8d_path,
                         "note": "the rasterizer's kernels (five with the direct bucket-rank binning, six with a scatter launch, seven with segment sort + merge): model bytes (every array once, DESIGN.md 4) over the sum of "
                                 "their launch durations; SURVEY.md 8d's figure (per-instance atomics charged) beside it"
                                 + ("; the deformation MLP's GEMMs are MFMA work: roofline_mlp" if deform else "")},
            "roofline_check": {"all_fractions_le_1": not violations, "violations": violations or None},
            "stages_ms": stages,
            "stages_note": "hipEvent pairs around each launch (mgs_set_option('profile', 2)): a pair reads ~2 us longer than "
                           "rocprofv3's kernel duration for the same launch, so their sum exceeds ms_per_step; the committed "
                           "profiles/r06_rocprofv3_kernel_stats_*.csv carry the kernel durations",
        }
        if not args.no_cpu_baseline and n_gpus == 1:
            if deform:
                out["cpu_baseline"] = cpu_baseline_dynamic(syn, sc, cam, d_color_h, d_feat_h, P, cfg.get("timesteps", 1),
                                                           renders_total, args.cpu_seconds * 2)
            else:
                out["cpu_baseline"] = cpu_baseline(syn, sc, cam, d_color_h, d_feat_h, P, args.cpu_seconds)
                # SURVEY.md 8d's protocol: the same port on 8 threads and on 1 beside all the host's cores (bounded: ~10 s each)
                out["cpu_baseline_threads8"] = cpu_baseline(syn, sc, cam, d_color_h, d_feat_h, P, args.cpu_seconds * 0.6,
                                                            threads=8, max_passes=3)
                out["cpu_baseline_threads1"] = cpu_baseline(syn, sc, cam, d_color_h, d_feat_h, P, args.cpu_seconds * 0.6,
                                                            threads=1, max_passes=2)
            out["gpu_over_cpu"] = value / out["cpu_baseline"]["value"]
        else:
            out["cpu_baseline"] = None
        if n_gpus == 1 and not deform and not args.no_reference_kernels:  # (independent of --no-cpu-baseline: it runs on the GPU)
            out["reference_kernels_same_gpu"] = reference_kernels(syn, sc, cam, d_color_h, d_feat_h, P, F, value)
        # the LAST keys of the line are the ones a truncated tail still shows: what every mode measured, the package default
        # ("eager-safe": what an unmodified caller gets) first
        out["headline_mode"] = mode
        out["ms_per_step_package_default"] = out["modes_ms_per_step"].get("eager-safe")
        out["modes_ms_per_step"] = out.pop("modes_ms_per_step")
        print(json.dumps(out))
        if violations:
            print("bench.py: roofline fraction above 1: " + "; ".join(violations), file=sys.stderr)
    rc = 3 if (args.strict_roofline and rank == 0 and violations) else 0
    if coll_on:
        # Every rank has finished (rank 0 has printed).  The process then leaves WITHOUT tearing the process group down:
        # destroy_process_group() / interpreter exit with HIP graphs alive that captured RCCL kernels was seen to abort
        # intermittently in ProcessGroupNCCL's background threads (round 5, one-rank dry run inside the GPU suite) -- after
        # the result line, but a non-zero exit code is a failed run to a driver.  The OS reclaims everything.
        dist.barrier()
        torch.cuda.synchronize()
        sys.stdout.flush()
        sys.stderr.flush()
        os._exit(rc)
    if rc:
        raise SystemExit(rc)


if __name__ == "__main__":
    main()
