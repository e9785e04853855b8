"""World-size-2 gloo test of the multi-GPU data path: views sharded round-robin, each rank
back-propagates its views into the flat gradient bucket, ONE all-reduce, result == single-process sum over
all views.  The renderer here is the autograd Oracle A on CPU (tests may use the oracle); on the GPU the same
render_sharded() drives the HIP rasterizer (tests/test_gpu_parity.py)."""
import os
import socket
import sys
import types

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _make(P=150, V=4, W=32, H=32, F=3):
    from manigaussian_amd import synthetic as syn
    sc = syn.make_scene(P, F=F, M=4, seed=3)
    cams = syn.circle_cameras(V, W, H, negative_focal=True)
    items = []
    for v, cam in enumerate(cams):
        dC, dF = syn.make_cotangents(W, H, F, seed=10 + v)
        items.append((types.SimpleNamespace(**syn.camera_settings_kwargs(cam, 1, True)), dC, dF))
    return sc, items


def _render_item(params, item):
    from oracle import oracle_a
    st, dC, dF = item
    c, f, _, _ = oracle_a.rasterize(params["means3D"], params["opacities"], st, shs=params["shs"],
                                    language_feature=params["language_feature"], scales=params["scales"],
                                    rotations=params["rotations"])
    return (c * dC).sum() + (f * dF).sum()


def _worker(rank, world, port, out_path):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from manigaussian_amd.parallel import GradBucket, render_sharded, shard_indices
    sc, items = _make()
    params = {k: v.clone().requires_grad_(True) for k, v in sc.items()}
    bucket = GradBucket(params)
    losses, _ = render_sharded(params, items, _render_item, bucket, rank=rank, world=world)
    assert len(losses) == len(shard_indices(len(items), rank, world))
    # a second step must not accumulate on top of the first (bucket is re-zeroed)
    render_sharded(params, items, _render_item, bucket, rank=rank, world=world)
    if rank == 0:
        torch.save({k: p.grad.clone() for k, p in params.items()}, out_path)
    dist.barrier()
    dist.destroy_process_group()


def test_shard_indices():
    from manigaussian_amd.parallel import shard_indices
    assert shard_indices(16, 1, 4) == [1, 5, 9, 13]
    assert shard_indices(3, 3, 4) == []
    assert sorted(sum((shard_indices(8, r, 8) for r in range(8)), [])) == list(range(8))


@pytest.mark.timeout(300)
def test_two_rank_all_reduce_equals_single_process_sum(tmp_path):
    out = str(tmp_path / "grads.pt")
    mp.start_processes(_worker, args=(2, _free_port(), out), nprocs=2, join=True, start_method="spawn")
    got = torch.load(out)
    sys.path.insert(0, ROOT)
    from manigaussian_amd.parallel import GradBucket, render_sharded
    sc, items = _make()
    params = {k: v.clone().requires_grad_(True) for k, v in sc.items()}
    bucket = GradBucket(params)
    render_sharded(params, items, _render_item, bucket, rank=0, world=1)
    for k, p in params.items():
        ref = p.grad
        assert torch.allclose(got[k], ref, rtol=1e-5, atol=1e-6 * (ref.abs().max().item() + 1e-12)), k
        assert p.grad.data_ptr() == bucket.views[k].data_ptr()  # gradients alias the single flat bucket


def _worker_alias(rank, world, port, out_path):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from manigaussian_amd.parallel import all_reduce_grads, flat_alias
    flat = torch.arange(20, dtype=torch.float32) * (rank + 1)
    pad, a, b = flat.split_with_sizes([4, 6, 10])          # the shape the HIP backward hands out: views of one buffer
    assert flat_alias([a.view(2, 3), b.view(5, 2)]).data_ptr() == a.data_ptr()
    all_reduce_grads([a.view(2, 3), b.view(5, 2)])        # in place on the shared storage
    c, d = torch.ones(3) * (rank + 1), torch.ones(2, 2) * (rank + 1)
    assert flat_alias([c, d]) is None
    all_reduce_grads([c, d])                               # staging-bucket path
    if rank == 0:
        torch.save({"flat": flat, "c": c, "d": d}, out_path)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_all_reduce_grads_aliasing_and_staging(tmp_path):
    out = str(tmp_path / "ar.pt")
    mp.start_processes(_worker_alias, args=(2, _free_port(), out), nprocs=2, join=True, start_method="spawn")
    got = torch.load(out)
    base = torch.arange(20, dtype=torch.float32)
    expect = base * 3.0
    expect[:4] = base[:4]                                  # the leading scratch region is outside the reduced span
    assert torch.equal(got["flat"], expect)
    assert torch.equal(got["c"], torch.full((3,), 3.0)) and torch.equal(got["d"], torch.full((2, 2), 3.0))


def _sparse_worker(rank, world, port, out_path):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from manigaussian_amd.parallel import all_reduce_grads, sparse_all_reduce_grads
    P = 1000
    g = torch.Generator().manual_seed(100 + rank)
    visible = torch.rand(P, generator=g) < 0.4           # each rank saw another 40 % of the Gaussians
    shapes = [(P, 3), (P, 1), (P, 4, 3), (P, 3), (P, 4), (P, 32)]   # means3D, opacity, SH, scales, rotations, features
    grads = [torch.randn(*s, generator=g) * visible.reshape(-1, *([1] * (len(s) - 1))) for s in shapes]
    dense = [t.clone() for t in grads]
    all_reduce_grads(dense)
    K = sparse_all_reduce_grads(grads, visible)
    if rank == 0:
        torch.save(dict(K=K, visible=visible, sparse=grads, dense=dense), out_path)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_sparse_visible_row_reduce_equals_the_dense_sum(tmp_path):
    """parallel.sparse_all_reduce_grads: only rows some rank saw (radii > 0) travel; the result is the dense all-reduce's,
    bit for bit (two ranks: a + b in both), and K = |union of the visible sets|."""
    out = str(tmp_path / "sparse.pt")
    mp.start_processes(_sparse_worker, args=(2, _free_port(), out), nprocs=2, join=True, start_method="spawn")
    got = torch.load(out)
    v1 = torch.rand(1000, generator=torch.Generator().manual_seed(101)) < 0.4
    assert got["K"] == int((got["visible"] | v1).sum()) and 0 < got["K"] < 1000
    for a, b in zip(got["sparse"], got["dense"]):
        assert torch.equal(a, b)


# ---- the deformation MLP sharded by point (parallel.sharded_deformation) --------------------------------------------------------

def _torch_assemble(lat, z, xyz, sh, rot, scale, op, feat, action):
    """models_embed.py:258-287 in torch (the CPU stand-in for the HIP assembly kernel, which test_reference_modules.py pins
    bit for bit against the reference module)."""
    N = lat.shape[0]
    parts = [lat, xyz.detach(), sh.detach()[:, 0], sh.detach()[:, 1:].reshape(N, 9), rot.detach(), scale.detach(),
             op.detach()] + ([feat.detach()] if feat is not None else []) + [z]
    if action is not None:
        parts.append(action.repeat(N, 1))
    return torch.cat(parts, -1)


def _torch_apply(delta, xyz, rot):
    """models_embed.py:295-299."""
    return xyz.detach() + delta[:, :3], torch.nn.functional.normalize(rot.detach() + delta[:, 3:], dim=-1)


def _dyn_setup(P, T, V, W=24, H=24, F=3):
    from manigaussian_amd import synthetic as syn
    from manigaussian_amd.deform import DeformationField
    sc = syn.make_scene(P, F=F, M=4, seed=5)
    torch.manual_seed(11)
    field = DeformationField(d_latent=16, d_hidden=32)
    with torch.no_grad():
        for p_ in field.parameters():
            p_.mul_(0.3)
        for b in field.mlp.blocks:  # the reference zero-initialises fc_1 (resnetfc.py:26): give every weight a gradient path
            b.fc_1.weight.normal_(0, 0.05)
    g = torch.Generator().manual_seed(7)
    latent, zf = torch.randn(P, 16, generator=g), torch.randn(P, 39, generator=g)
    actions = [torch.randn(1, 8, generator=g) for _ in range(T)]
    cams = syn.circle_cameras(V, W, H, negative_focal=True)
    views = []
    for v, cam in enumerate(cams):
        dC, dF = syn.make_cotangents(W, H, F, seed=20 + v)
        views.append((types.SimpleNamespace(**syn.camera_settings_kwargs(cam, 1, True)), dC, dF))
    return sc, field, latent, zf, actions, views


def _dyn_step(plan, sc, field, latent_local, zf_local, actions, views, bucket):
    """One dynamic step on this rank: for each of its timesteps the (point-sharded) MLP, then Oracle A renders of its views."""
    from manigaussian_amd.parallel import sharded_deformation
    from oracle import oracle_a
    bucket.attach()
    for t in plan.timesteps:
        nxt = sharded_deformation(field, latent_local, zf_local, sc["means3D"], sc["shs"], sc["rotations"], sc["scales"],
                                  sc["opacities"], action=actions[t], group=plan.group, assemble=_torch_assemble,
                                  apply=_torch_apply)
        loss = 0.0
        for v in plan.views:
            st, dC, dF = views[v]
            c, f, _, _ = oracle_a.rasterize(nxt["xyz"], nxt["opacity"], st, shs=nxt["sh"],
                                            language_feature=sc["language_feature"], scales=nxt["scale"], rotations=nxt["rot"])
            loss = loss + (c * dC).sum() * (1.0 + 0.1 * t) + (f * dF).sum()
        loss.backward()
    return bucket.all_reduce()  # over the WORLD: every rank holds a share of every parameter's gradient


def _dyn_worker(rank, world, port, out_dir, P, T, V):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from manigaussian_amd.parallel import DynamicPlan, GradBucket
    sc, field, latent, zf, actions, views = _dyn_setup(P, T, V)
    plan = DynamicPlan(T, V, rank, world)
    lo, hi = plan.point_rows(P)
    latent_local = latent[lo:hi].clone().requires_grad_(True)   # a LOCAL leaf: no collective for its gradient
    bucket = GradBucket(dict(field.named_parameters()))
    _dyn_step(plan, sc, field, latent_local, zf[lo:hi], actions, views, bucket)
    torch.save(dict(lo=lo, hi=hi, timesteps=plan.timesteps, views=plan.views, latent_grad=latent_local.grad,
                    world_size_seen=dist.get_world_size(), group_size=plan.group_size,
                    params={k: p.grad.clone() for k, p in field.named_parameters()}, desc=plan.describe(P)),
               os.path.join(out_dir, f"rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(900)
@pytest.mark.parametrize("world,T,V,P", [(2, 1, 4, 151), (4, 2, 2, 96), (2, 2, 2, 80), (8, 1, 8, 93), (8, 4, 4, 70)],
                         ids=["c5-like:2ranks-share-1-timestep", "4ranks-2groups", "c4-like:1-timestep-per-rank",
                              "configs[4]-on-8:all-8-ranks-share-the-timestep-P-not-divisible",
                              "configs[3]-on-8:2-ranks-per-timestep-2-views-each"])
def test_point_sharded_deformation_equals_the_single_process_step(tmp_path, world, T, V, P):
    """parallel.sharded_deformation + DynamicPlan: ranks that share a timestep split its MLP by point (all-gather of the
    deltas, reduce-scatter of dL/d delta), render their views, all-reduce the MLP-gradient bucket.  Every MLP parameter's
    gradient and every row of point_latent's must equal the single-process step's (all timesteps, all views, no sharding)
    to 1e-5 of the tensor's max; a row count that does not divide (151 over 2) exercises the padded blocks."""
    mp.start_processes(_dyn_worker, args=(world, _free_port(), str(tmp_path), P, T, V), nprocs=world, join=True,
                       start_method="spawn")
    sys.path.insert(0, ROOT)
    from manigaussian_amd.parallel import DynamicPlan, GradBucket
    sc, field, latent, zf, actions, views = _dyn_setup(P, T, V)
    lat = latent.clone().requires_grad_(True)
    bucket = GradBucket(dict(field.named_parameters()))
    _dyn_step(DynamicPlan(T, V, 0, 1), sc, field, lat, zf, actions, views, bucket)
    ref_params = {k: p.grad for k, p in field.named_parameters()}
    got = [torch.load(os.path.join(str(tmp_path), f"rank{r}.pt")) for r in range(world)]
    covered = torch.zeros(T, P)
    views_done = torch.zeros(T, V)
    for r, g in enumerate(got):
        assert g["world_size_seen"] == world and g["group_size"] == max(1, world // T)
        for t in g["timesteps"]:
            views_done[t, g["views"]] += 1
        for k, ref in ref_params.items():
            scale = ref.abs().max().item()
            assert scale > 0, f"{k}: the reference gradient is all zero -- the test would prove nothing"
            assert (g["params"][k] - ref).abs().max().item() <= 1e-5 * scale, (r, k)
        for t in g["timesteps"]:
            covered[t, g["lo"]:g["hi"]] += 1
        if world > T:  # the rank's rows of point_latent carry the complete gradient of ITS timestep ... (see below)
            assert g["desc"]["mlp_points_per_rank"] == g["hi"] - g["lo"] < P
            assert g["desc"]["all_gather_bytes_per_timestep"] == 28 * P
    assert torch.equal(covered, torch.ones(T, P)), "every (timestep, point) must be evaluated by exactly one rank"
    assert torch.equal(views_done, torch.ones(T, V)), "every (timestep, view) must be rendered by exactly one rank"
    # point_latent: summed over the timesteps in the single-process step; a rank's local leaf holds its timesteps' share of
    # its rows, so the ranks' contributions add up to the reference row by row
    total = torch.zeros_like(lat.grad)
    for g in got:
        total[g["lo"]:g["hi"]] += g["latent_grad"]
    assert (total - lat.grad).abs().max().item() <= 1e-5 * lat.grad.abs().max().item()


# ---- bench.py --gpus N: the driver's command shape must launch N ranks by itself ------------------------------------------

def _bench(args, timeout=600):
    import json
    import subprocess
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True,
                       timeout=timeout, env={k: v for k, v in os.environ.items()
                                             if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")})
    line = next((ln for ln in reversed(r.stdout.splitlines()) if ln.startswith("{")), None)
    return r, (json.loads(line) if line else None)


def test_bench_refuses_to_run_fewer_ranks_than_asked():
    """`python bench.py --gpus 2` must never print an n_gpus = 1 line: without two HIP devices it exits with an error
    (here: no device at all), and under a launcher it insists on WORLD_SIZE == --gpus."""
    if torch.cuda.is_available() and torch.cuda.device_count() >= 2:
        pytest.skip("two devices present: the real launch is covered by the gpu test below")
    r, j = _bench(["--gpus", "2", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"], timeout=300)
    assert r.returncode != 0 and j is None
    assert "HIP device" in (r.stderr + r.stdout)
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    import subprocess
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], capture_output=True, text=True,
                       timeout=300, env=env)
    assert r.returncode != 0 and "WORLD_SIZE" in (r.stderr + r.stdout)


@pytest.mark.gpu
@pytest.mark.timeout(600)
def test_bench_self_launches_two_ranks_on_one_device():
    """The N > 1 path end to end where only one GPU exists: `bench.py --gpus 2` re-executes itself under
    torch.distributed.run; --one-device --backend gloo are the testing flags (both ranks on cuda:0).  The JSON line must
    say 2 ranks, name both devices, and carry the all-reduce volume (the parameter gradients only: 55 floats/Gaussian)."""
    r, j = _bench(["--gpus", "2", "--one-device", "--backend", "gloo", "--steps", "20", "--warmup", "5", "--mode", "eager-st",
                   "--only-mode", "--no-cpu-baseline", "--P", "20000"])
    assert r.returncode == 0 and j is not None, (r.stdout[-2000:], r.stderr[-3000:])
    assert j["n_gpus"] == 2 and j["distributed"]["world_size_seen"] == 2 and len(j["distributed"]["device_ids"]) == 2
    assert j["distributed"]["allreduce_bytes_per_step"] == 20000 * (3 + 1 + 12 + 3 + 4 + 32) * 4
    assert j["distributed"]["allreduce_exposed_ms_per_step"] is not None and j["value"] > 0


@pytest.mark.gpu
@pytest.mark.timeout(900)
@pytest.mark.parametrize("cfg,ranks", [("c4", 1), ("c4", 2), ("c5", 2), ("c4", 8), ("c5", 8)])
def test_bench_dynamic_configs_share_the_step_between_ranks(cfg, ranks):
    """BASELINE configs[3] / [4] as `bench.py --config c4 | c5` runs them (at a reduced Gaussian count, the shapes and the
    control flow are the real ones): the step's (timestep, view) renders are SHARED by the ranks (strong scaling), every
    rank reduces exactly the flat MLP-gradient bucket, and the line says so."""
    env_flags = ["--one-device", "--backend", "gloo"] if ranks > 1 else []
    os.environ["MGS_NO_GEMM_TUNING"] = "1"
    try:
        r, j = _bench(["--config", cfg, "--gpus", str(ranks), "--P", "6000", "--steps", "3", "--warmup", "2", "--mode",
                       "eager-st", "--only-mode", "--no-cpu-baseline"] + env_flags, timeout=800)
    finally:
        del os.environ["MGS_NO_GEMM_TUNING"]
    assert r.returncode == 0 and j is not None, (r.stdout[-2000:], r.stderr[-3000:])
    total = {"c4": 16, "c5": 8}[cfg]
    c = j["config"]
    assert j["n_gpus"] == ranks and j["scaling"] == "strong" and j["steps"] == 3
    assert c["renders_per_step_total"] == total and c["renders_per_step_per_gpu"] == total // ranks
    assert j["value"] > 0 and abs(j["value"] - 6000 * total * 3 / (j["ms_per_step"] * 3e-3)) <= 1e-6 * j["value"]
    if ranks > 1:
        from manigaussian_amd.deform import DeformationField
        n_params = sum(p.numel() for p in DeformationField().parameters())
        assert j["distributed"]["allreduce_bytes_per_step"] == 4 * n_params  # the MLP's gradients, nothing else
        assert j["distributed"]["world_size_seen"] == ranks and len(j["distributed"]["device_ids"]) == ranks
        part = c["partition"]
        # the driver's 8-GPU shapes: configs[3] = 4 timesteps x 4 views (2 ranks share a timestep, 2 views each),
        # configs[4] = 1 timestep x 8 views (all 8 ranks share it, one view each, 6000 / 8 = 750 points of its MLP each)
        T_total = {"c4": 4, "c5": 1}[cfg]
        assert part["ranks_sharing_a_timestep"] == max(1, ranks // T_total)
        assert part["mlp_points_per_rank"] == -(-6000 // max(1, ranks // T_total))


@pytest.mark.gpu
@pytest.mark.timeout(900)
def test_bench_two_ranks_over_rccl():
    """Two MI355X, RCCL: skipped on a one-GPU box (the driver's scaling run covers 1/2/4/8)."""
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two HIP devices")
    r, j = _bench(["--gpus", "2", "--steps", "50", "--warmup", "10", "--no-cpu-baseline"])
    assert r.returncode == 0 and j is not None, (r.stdout[-2000:], r.stderr[-3000:])
    assert j["n_gpus"] == 2 and j["distributed"]["backend"] == "nccl" and sorted(j["distributed"]["device_ids"]) == [0, 1]


@pytest.mark.gpu
@pytest.mark.timeout(900)
@pytest.mark.parametrize("cfg", ["c3", "c4"])
def test_rccl_call_sequence_runs_on_one_device(cfg):
    """the RCCL branches of the N > 1 path (parallel.py: the in-place flat-bucket all-reduce with two
    alternating buffers, DynamicPlan's dist.new_group, all_gather_into_tensor / reduce_scatter_tensor of DEVICE tensors and
    their autograd glue) had only ever run over gloo through host staging.  `bench.py --dry-collectives` creates an RCCL
    process group of ONE rank and issues the exact call sequence to the device; not skipped on a one-GPU box.  What this
    proves: the calls are well-formed for RCCL on this stack (dtypes, contiguity, sizes, stream semantics, sub-groups); what
    it cannot: transport over xGMI."""
    args = ["--dry-collectives", "--steps", "4", "--warmup", "2", "--no-cpu-baseline", "--no-reference-kernels"]
    if cfg == "c3":
        args += ["--P", "20000", "--mode", "graph"]  # all modes: graph replay with alternating buckets, eager, ...
    else:
        args += ["--config", "c4", "--P", "6000", "--mode", "eager-st", "--only-mode"]
        os.environ["MGS_NO_GEMM_TUNING"] = "1"
    try:
        r, j = _bench(args, timeout=800)
    finally:
        os.environ.pop("MGS_NO_GEMM_TUNING", None)
    assert r.returncode == 0 and j is not None, (r.stdout[-2000:], r.stderr[-3000:])
    d = j["distributed"]
    assert d["dry_collectives"] is True and d["backend"] == "nccl" and d["world_size_seen"] == 1 and j["n_gpus"] == 1
    assert d["allreduce_exposed_ms_per_step"] is not None and d["allreduce_bytes_per_step"] > 0
    if cfg == "c3":
        assert d["allreduce_bytes_per_step"] == 20000 * (3 + 1 + 12 + 3 + 4 + 32) * 4
        assert not j.get("mode_errors"), j.get("mode_errors")
    else:
        part = j["config"]["partition"]
        assert part["all_gather_bytes_per_timestep"] == 28 * 6000 and part["reduce_scatter_bytes_per_timestep"] == 28 * 6000
