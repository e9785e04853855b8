"""Multi-GPU data path: independent (view, timestep) renders sharded over ranks, one all-reduce of the
per-Gaussian gradients.

One process per GPU, torch.distributed over RCCL (backend "nccl" on ROCm) / gloo on CPU.  The path has no
exchange step inside forward or backward -- every render reads the replicated Gaussian set and adds its
gradient contribution -- so the only collective is ONE sum all-reduce per step over a single flat fp32
bucket that all gradient tensors alias (no per-tensor launches, no copy into a staging buffer).

The reference reaches multi-GPU only through Lightning Fabric DDP over replay samples
(train.py:94-95, agents/manigaussian_bc/qattention_manigaussian_bc_agent.py:155,163) and renders strictly
one view per call (agents/manigaussian_bc/neural_rendering.py:386).
"""
from typing import Callable, Dict, List, Optional, Sequence

import torch
import torch.distributed as dist


# Dry run of the collectives (bench.py --dry-collectives, tests/test_parallel.py): with a process group of ONE rank every
# collective below is the identity and is skipped -- unless this flag is set, in which case the exact call sequence of the N > 1
# path (sub-group creation, padded all-gather / reduce-scatter of device tensors, the flat-bucket all-reduce) is issued to the
# backend anyway.  On a single MI355X that runs the RCCL branches on hardware; it measures the calls' cost, not transport.
_DRY = [False]


def set_dry_collectives(on: bool) -> bool:
    old, _DRY[0] = _DRY[0], bool(on)
    return old


def collectives_active(group=None) -> bool:
    """Is there somebody to talk to (a process group of more than one rank -- or a dry run)?"""
    if not (dist.is_available() and dist.is_initialized()):
        return False
    return _DRY[0] or dist.get_world_size(group) > 1


def shard_indices(n_items: int, rank: int, world: int) -> List[int]:
    """Round-robin: rank g takes items {i : i mod world == g}."""
    return list(range(rank, n_items, world))


class GradBucket:
    """One flat fp32 buffer; each parameter's .grad is a view into it."""

    def __init__(self, params: Dict[str, torch.Tensor]):
        self.names = list(params)
        self.params = params
        dev = next(iter(params.values())).device
        sizes = [params[n].numel() for n in self.names]
        self.flat = torch.zeros(sum(sizes), dtype=torch.float32, device=dev)
        self.views = {}
        off = 0
        for n, sz in zip(self.names, sizes):
            self.views[n] = self.flat[off:off + sz].view_as(params[n])
            off += sz

    def attach(self):
        """Zero the bucket and (re)point every .grad at its view; autograd then accumulates in place."""
        self.flat.zero_()
        for n in self.names:
            self.params[n].grad = self.views[n]

    def all_reduce(self, group=None, async_op: bool = False):
        if collectives_active(group):
            return dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=group, async_op=async_op)
        return None


def render_sharded(params: Dict[str, torch.Tensor], items: Sequence, render_item: Callable, bucket: GradBucket,
                   rank: int = 0, world: int = 1, group=None):
    """Render this rank's share of `items` (views / (timestep, view) pairs), back-propagate each one into
    the shared bucket, then all-reduce the bucket once.

    render_item(params, item) -> scalar loss tensor whose backward reaches `params`.
    Returns (local losses, work handle or None)."""
    bucket.attach()
    losses = []
    for i in shard_indices(len(items), rank, world):
        loss = render_item(params, items[i])
        loss.backward()
        losses.append(loss.detach())
    handle = bucket.all_reduce(group)
    return losses, handle


def flat_alias(grads: Sequence[torch.Tensor]) -> Optional[torch.Tensor]:
    """If every tensor in `grads` is a dense view into ONE storage (the HIP backward hands all gradients out
    of a single allocation, manigaussian_amd/_C.py), return a 1-D tensor aliasing the span they cover so
    that one collective reaches all of them in place; else None."""
    gs = [g for g in grads if g is not None and g.numel() > 0]
    if not gs:
        return None
    st = gs[0].untyped_storage()
    base_ptr = st.data_ptr()
    lo, hi = None, None
    for g in gs:
        if g.dtype != torch.float32 or not g.is_contiguous() or g.untyped_storage().data_ptr() != base_ptr:
            return None
        b = g.storage_offset()
        lo = b if lo is None else min(lo, b)
        hi = b + g.numel() if hi is None else max(hi, b + g.numel())
    return torch.empty(0, dtype=torch.float32, device=gs[0].device).set_(st, lo, (hi - lo,), (1,))


def all_reduce_grads(grads: Sequence[torch.Tensor], group=None, async_op: bool = False):
    """ONE sum all-reduce over the gradients of a step.  In place on the shared allocation when the gradients
    alias one buffer (no staging copy); otherwise through a flat staging bucket that is scattered back."""
    if not collectives_active(group):
        return None
    flat = flat_alias(grads)
    if flat is not None:
        return dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group, async_op=async_op)
    gs = [g for g in grads if g is not None]
    bucket = torch.cat([g.reshape(-1) for g in gs])
    dist.all_reduce(bucket, op=dist.ReduceOp.SUM, group=group)  # blocking: the scatter-back below needs the result
    off = 0
    for g in gs:
        g.copy_(bucket[off:off + g.numel()].view_as(g))
        off += g.numel()
    return None


def sparse_all_reduce_grads(grads: Sequence[torch.Tensor], visible: torch.Tensor, group=None,
                            check_rows: bool = False) -> int:
    """The same sum as all_reduce_grads, moving only the rows that can be non-zero (lever 2 of DESIGN.md 7).

    grads:   per-Gaussian gradient tensors of one step, each [P, ...] (means3D, opacity, SH, scales, rotations, features ...)
    visible: bool [P] on this rank -- radii > 0 in ANY view this rank rendered.  A Gaussian that no rank saw has an all-zero
             gradient row on every rank (the backward writes zeros for radii == 0, RAST/rasterize_points.cu:167-184 zero
             initialises them), so leaving those rows out of the reduction is exact.
    Cost model: one small all-reduce of the mask (4 P bytes), one host read of the row count (the ranks must agree on the
    packed size), a gather / scatter pass over the visible rows, and an all-reduce of K x D floats instead of P x D.  At
    BASELINE configs[2] about 60 % of the Gaussians are visible per view; the saving grows with the number of ranks (ring
    all-reduce is per-link bound on xGMI) and shrinks with the number of views per rank.
    Contract: ONLY gradients that are exactly zero outside the union of the ranks' `visible` rows -- the rasterizer's
    per-Gaussian gradients; not weight decay, not anything that flowed through an MLP (check_rows=True verifies it, at the
    price of a pass over the gradients and a host read).  Returns K = |union of the visible sets| (the local visible count
    without a process group: the union over one rank)."""
    if not collectives_active(group):
        return int(visible.sum().item())
    gs = [g for g in grads if g is not None and g.numel() > 0]
    P = int(visible.numel())
    for g in gs:
        if g.shape[0] != P or not g.is_contiguous():
            raise ValueError("sparse_all_reduce_grads takes contiguous per-Gaussian tensors [P, ...]")
        if g.dtype != gs[0].dtype or g.device != gs[0].device:
            raise ValueError("sparse_all_reduce_grads: every gradient must have one dtype and live on one device")
    if check_rows:  # debug: a non-zero row outside `visible` would be left un-reduced (rasterizer gradients never have one)
        hidden = ~visible.to(torch.bool)
        for g in gs:
            if bool((g.view(P, -1)[hidden] != 0).any()):
                raise ValueError("sparse_all_reduce_grads: a gradient has a non-zero row for a Gaussian this rank did not "
                                 "see -- only per-Gaussian gradients of the rasterizer qualify (no weight decay / MLP terms)")
    seen = visible.to(torch.int32)
    dist.all_reduce(seen, op=dist.ReduceOp.SUM, group=group)   # union over the ranks
    idx = (seen > 0).nonzero(as_tuple=False).squeeze(1)        # host synchronisation: every rank learns the same K
    K = int(idx.numel())
    if K == 0:
        return 0
    rows = [g.view(P, -1) for g in gs]
    packed = torch.cat([r.index_select(0, idx) for r in rows], dim=1)
    dist.all_reduce(packed, op=dist.ReduceOp.SUM, group=group)
    off = 0
    for r in rows:
        w = r.shape[1]
        r.index_copy_(0, idx, packed[:, off:off + w])
        off += w
    return K


# ---- the deformation MLP sharded BY POINT (BASELINE configs[3] / [4]) ------------------------------------------
#
# Views shard the RENDERS of a step, but the per-Gaussian deformation MLP in front of them (agents/manigaussian_bc/
# models_embed.py:256-304 -> resnetfc.py:137-177) is evaluated once per timestep over ALL points, whatever the number of
# views: with one timestep x 8 views on 8 GPUs (configs[4]) every rank would run the whole 500 000-point MLP (72 ms) to render
# one view (0.5 ms).  The reference has no other choice -- its only multi-GPU axis is DDP over replay samples (train.py:94-95,
# neural_rendering.py:386 `assert bs == 1`) -- but the MLP is independent per point, so here the ranks that share a timestep
# split its POINTS:
#
#   rank r: delta[lo_r:hi_r] = MLP(inputs[lo_r:hi_r])                          P / G points each
#   all-gather   delta [P, 7]            (28 P bytes; 14 MB at 500 000 points)   -> every rank applies and renders ITS views
#   ... render backward on every rank yields dL/d delta [P, 7] for its views ...
#   reduce-scatter (sum) dL/d delta      (28 P bytes)                            -> rank r back-propagates rows [lo_r, hi_r)
#   all-reduce of the flat MLP-gradient bucket (as before; each rank's bucket now holds its points' share)
#
# The gradient of point_latent needs no collective: rank r owns rows [lo_r, hi_r) and they are complete after the
# reduce-scatter (the sum over every rank's views).

def point_shard(n_points: int, rank: int, world: int):
    """(lo, hi, per): rank's contiguous rows [lo, hi) of n_points, per = rows per rank the collectives are padded to."""
    per = (n_points + world - 1) // world
    lo = min(n_points, rank * per)
    return lo, min(n_points, lo + per), per


def _staged(t: torch.Tensor, group) -> bool:
    """gloo has no device-tensor all-gather / reduce-scatter: the test-only combination (two ranks on ONE GPU over gloo,
    bench.py --one-device --backend gloo) stages through host memory.  RCCL ("nccl") never does."""
    return t.is_cuda and dist.get_backend(group) == "gloo"


class _GatherRows(torch.autograd.Function):
    """forward: all-gather of every rank's row block -> the full [n_points, C] tensor on every rank;
    backward: reduce-scatter (sum over ranks) of dL/d(full) -> this rank's rows."""

    @staticmethod
    def forward(ctx, local, n_points, group):
        world, rank = dist.get_world_size(group), dist.get_rank(group)
        lo, hi, per = point_shard(n_points, rank, world)
        assert local.shape[0] == hi - lo, f"rank {rank} holds rows [{lo}, {hi}) but was given {local.shape[0]}"
        C = local.shape[1]
        send = local.contiguous()
        if hi - lo != per:  # the last rank(s): pad to the common block size
            send = torch.cat([send, send.new_zeros(per - (hi - lo), C)])
        full = send.new_empty(world * per, C)
        if _staged(send, group):
            host = torch.empty(world * per, C, dtype=send.dtype)
            dist.all_gather_into_tensor(host, send.cpu(), group=group)
            full.copy_(host)
        else:
            dist.all_gather_into_tensor(full, send, group=group)
        ctx.meta = (n_points, group, lo, hi, per, world)
        return full[:n_points]

    @staticmethod
    def backward(ctx, g_full):
        n_points, group, lo, hi, per, world = ctx.meta
        g = g_full.contiguous()
        if g.shape[0] != world * per:
            g = torch.cat([g, g.new_zeros(world * per - g.shape[0], g.shape[1])])
        out = g.new_empty(per, g.shape[1])
        if _staged(g, group):
            host = torch.empty(per, g.shape[1], dtype=g.dtype)
            dist.reduce_scatter_tensor(host, g.cpu(), op=dist.ReduceOp.SUM, group=group)
            out.copy_(host)
        else:
            dist.reduce_scatter_tensor(out, g, op=dist.ReduceOp.SUM, group=group)
        return out[:hi - lo], None, None


def gather_rows(local: torch.Tensor, n_points: int, group=None) -> torch.Tensor:
    """Differentiable all-gather of per-point rows (see _GatherRows); the identity without a process group of > 1 ranks."""
    if not collectives_active(group):
        return local
    return _GatherRows.apply(local, n_points, group)


def sharded_deformation(field, point_latent, z_feature, xyz, sh, rot, scale, opacity, feature=None, action=None, *,
                        group=None, assemble=None, apply=None):
    """DeformationField.forward (models_embed.py:255-304) with the MLP evaluated on THIS rank's points only.

    point_latent / z_feature: either this rank's rows [hi - lo, .] (a local leaf: its .grad is complete after backward, no
    collective) or the full [P, .] tensors (sliced here).  xyz, sh, rot, scale, opacity (feature): the full, replicated
    Gaussian set -- they are detached in the reference and every rank needs all of them to render.
    group: the process group whose ranks split the points (DynamicPlan.group: dist.group.WORLD or a sub-group); None = no
    sharding, the plain DeformationField.forward on this rank (NOT torch's "None means the world").
    assemble / apply: the input-assembly and apply operators (default: the HIP kernels of manigaussian_amd.deform; the CPU
    tests hand in torch restatements).  Returns the dict DeformationField.forward returns, identical on every rank of `group`.
    """
    from . import deform
    assemble = assemble or deform.assemble_deform_input
    apply = apply or deform.deform_apply
    P = xyz.shape[0]
    sharded = group is not None and collectives_active(group)
    lo, hi, _ = point_shard(P, dist.get_rank(group), dist.get_world_size(group)) if sharded else (0, P, P)

    def rows(t, local_ok=False):
        if t is None:
            return None
        if local_ok and t.shape[0] == hi - lo:
            return t
        assert t.shape[0] == P, f"expected {P} (or this rank's {hi - lo}) rows, got {t.shape[0]}"
        return t if (lo == 0 and hi == P) else t[lo:hi]

    zx = assemble(rows(point_latent, True), rows(z_feature, True), rows(xyz), rows(sh), rows(rot), rows(scale),
                  rows(opacity), rows(feature) if getattr(field, "use_semantic_feature", False) else None,
                  action if getattr(field, "use_action", True) else None)
    delta_local, _ = field.mlp(zx)
    delta = gather_rows(delta_local, P, group) if sharded else delta_local
    next_xyz, next_rot = apply(delta, xyz, rot)
    return dict(xyz=next_xyz, rot=next_rot, sh=sh.detach(), scale=scale.detach(), opacity=opacity.detach(),
                feature=None if feature is None else feature.detach())


def timestep_groups(n_timesteps: int, rank: int, world: int):
    """How a dynamic step's T timesteps x V views are shared (strong scaling).  Returns (my_timesteps, group_rank, group_size,
    group_ranks):
      world <= T: rank r evaluates timesteps r, r + world, ... alone (the MLP work is already split by timestep);
      world  > T: the world is cut into T groups of world / T consecutive ranks; a group shares ONE timestep -- its MLP by
                  point (sharded_deformation over the group), its views round-robin (group_rank, group_size).
    world must divide T or be a multiple of it."""
    if world <= n_timesteps:
        if n_timesteps % world:
            raise ValueError(f"{n_timesteps} timesteps do not divide over {world} ranks")
        return list(range(rank, n_timesteps, world)), 0, 1, [rank]
    if world % n_timesteps:
        raise ValueError(f"{world} ranks are not a multiple of {n_timesteps} timesteps")
    gsz = world // n_timesteps
    t = rank // gsz
    return [t], rank % gsz, gsz, list(range(t * gsz, (t + 1) * gsz))


class DynamicPlan:
    """This rank's share of a dynamic step of T timesteps x V views (BASELINE configs[3] / [4]; strong scaling).

    timesteps   the timesteps whose MLP this rank evaluates (wholly, or its point rows of)
    views       the views of those timesteps this rank renders
    group       the process group that shares this rank's timestep (None: the rank works alone on its timesteps)
    group_rank, group_size
    """

    def __init__(self, n_timesteps: int, n_views: int, rank: int = 0, world: int = 1):
        self.n_timesteps, self.n_views, self.rank, self.world = n_timesteps, n_views, rank, world
        self.timesteps, self.group_rank, self.group_size, self.group_ranks = timestep_groups(n_timesteps, rank, world)
        if n_views % self.group_size:
            raise ValueError(f"{n_views} views do not divide over the {self.group_size} ranks that share a timestep")
        self.views = list(range(self.group_rank, n_views, self.group_size))
        self.group = None
        if _DRY[0] and self.group_size == 1 and dist.is_available() and dist.is_initialized():
            # dry run: a sub-group of this one rank stands in for "the ranks that share my timestep" (new_group, the padded
            # all-gather / reduce-scatter and their autograd glue all execute; the plan's shares stay those of one rank)
            self.group = dist.new_group([rank])
        if self.group_size > 1:
            if self.group_size == world:
                self.group = dist.group.WORLD
            else:  # every rank creates every group, in the same order (torch.distributed's rule)
                gsz = self.group_size
                for t in range(n_timesteps):
                    g = dist.new_group(list(range(t * gsz, (t + 1) * gsz)))
                    if t == self.timesteps[0]:
                        self.group = g

    def point_rows(self, n_points: int):
        """Rows [lo, hi) of the per-point MLP inputs this rank evaluates."""
        lo, hi, _ = point_shard(n_points, self.group_rank, self.group_size)
        return lo, hi

    def describe(self, n_points: int) -> dict:
        lo, hi = self.point_rows(n_points)
        sharded = self.group_size > 1 or self.group is not None  # (a dry run's one-rank sub-group issues them too)
        return {"timesteps_on_this_rank": len(self.timesteps), "views_per_timestep_on_this_rank": len(self.views),
                "ranks_sharing_a_timestep": self.group_size, "mlp_points_per_rank": hi - lo,
                "all_gather_bytes_per_timestep": 28 * n_points if sharded else 0,
                "reduce_scatter_bytes_per_timestep": 28 * n_points if sharded else 0}
