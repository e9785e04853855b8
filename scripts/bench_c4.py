"""BASELINE configs[3] in one process: 100k Gaussians + deformation MLP (d_in 73 -> 512 x 5 -> 7), timesteps x views renders.

  python scripts/bench_c4.py [--timesteps 1] [--views 4] [--steps 20] [--bf16]
  torchrun --nproc-per-node G scripts/bench_c4.py ...     (one timestep per rank; one all-reduce of the MLP gradients)

Per timestep: assemble the deformation input (HIP), run the MLP (torch GEMMs on hipBLASLt/MFMA), apply the deltas (HIP),
rasterize `views` views of the deformed set in ONE batched call (HIP), back-propagate an L2 image loss into the MLP and
the point latents.  Reports ms per step and the split MLP / rasterizer (hipEvents)."""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist
from manigaussian_amd import GaussianRasterizationSettings, GaussianRasterizerBatch
from manigaussian_amd import synthetic as syn
from manigaussian_amd.deform import DeformationField

ap = argparse.ArgumentParser()
ap.add_argument("--P", type=int, default=100000)
ap.add_argument("--timesteps", type=int, default=1, help="timesteps handled by THIS process")
ap.add_argument("--views", type=int, default=4)
ap.add_argument("--steps", type=int, default=20)
ap.add_argument("--F", type=int, default=32)
ap.add_argument("--bf16", action="store_true", help="autocast the MLP GEMMs to bf16 (rasterizer stays fp32)")
ap.add_argument("--no-tune", action="store_true", help="keep hipBLASLt's heuristic GEMM picks (default: TunableOp, deform.tune_gemms)")
args = ap.parse_args()
rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
local = int(os.environ.get("LOCAL_RANK", "0"))
torch.cuda.set_device(local)
dev = torch.device("cuda", local)
if world > 1:
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group("nccl", device_id=dev)
torch.autograd.set_multithreading_enabled(False)
P, V, W, F = args.P, args.views, 128, args.F
sc = {k: v.to(dev) for k, v in syn.make_scene(P, F=F, M=4, seed=0).items()}
g = torch.Generator().manual_seed(3)
point_latent = torch.randn(P, 128, generator=g).to(dev).requires_grad_(True)
z_feature = torch.randn(P, 39, generator=g).to(dev)
if not args.no_tune:
    from manigaussian_amd.deform import tune_gemms
    tune_gemms()
field = DeformationField().to(dev)
with torch.no_grad():  # the reference zero-initialises fc_1; give the deltas some life without exploding the scene
    for p_ in field.parameters():
        p_.mul_(0.05)
params = [p_ for p_ in field.parameters()] + [point_latent]
cams = syn.circle_cameras(max(V, 8), W, W, negative_focal=True)[:V]
rast = GaussianRasterizerBatch([GaussianRasterizationSettings(**syn.camera_settings_kwargs(c, 1, True, device=dev)) for c in cams])
targets = [(torch.rand(V, 3, W, W, generator=g).to(dev), torch.randn(V, F, W, W, generator=g).to(dev)) for _ in range(args.timesteps)]
actions = [torch.randn(1, 8, generator=g).to(dev) for _ in range(args.timesteps)]
ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]


def step(timed=False):
    total = 0.0
    for t in range(args.timesteps):
        if timed: ev[0].record()
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=args.bf16):
            nxt = field(point_latent, z_feature, sc["means3D"], sc["shs"], sc["rotations"], sc["scales"], sc["opacities"],
                        action=actions[t])
        if timed: ev[1].record()
        color, feat, _ = rast(nxt["xyz"].float(), None, nxt["opacity"], shs=nxt["sh"], language_feature_precomp=sc["language_feature"],
                              scales=nxt["scale"], rotations=nxt["rot"].float())
        loss = ((color - targets[t][0]) ** 2).mean() + 0.01 * ((feat - targets[t][1]) ** 2).mean()
        grads = torch.autograd.grad(loss, params)
        if timed: ev[2].record()
        total = total + loss.detach()
    if world > 1:  # one all-reduce of the flattened MLP gradients (22.9 MB), as DDP would do
        flat = torch.cat([g_.reshape(-1) for g_ in grads[:-1]])
        dist.all_reduce(flat)
    return total


for _ in range(5):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(args.steps):
    step()
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / args.steps * 1e3
step(timed=True)
torch.cuda.synchronize()
mlp_fwd = ev[0].elapsed_time(ev[1])
rest = ev[1].elapsed_time(ev[2])
if rank == 0:
    renders = args.timesteps * V * world
    print(f"C4: P={P} F={F} timesteps/proc={args.timesteps} views={V} world={world} bf16={args.bf16}: {ms:.3f} ms/step, "
          f"{renders} renders/step -> {P * renders / ms / 1e3:.0f} M Gaussians/s | last timestep: deformation fwd {mlp_fwd:.3f} ms, "
          f"raster fwd+bwd + MLP bwd {rest:.3f} ms")
if world > 1:
    dist.destroy_process_group()
