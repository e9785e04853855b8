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
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
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
    if not (dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1):
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


def sparse_all_reduce_grads(grads: Sequence[torch.Tensor], visible: torch.Tensor, group=None) -> int:
    """The same sum as all_reduce_grads, moving only the rows that can be non-zero (lever 2 of DESIGN.md 7).

    grads:   per-Gaussian gradient tensors of one step, each [P, ...] (means3D, opacity, SH, scales, rotations, features ...)
    visible: bool [P] on this rank -- radii > 0 in ANY view this rank rendered.  A Gaussian that no rank saw has an all-zero
             gradient row on every rank (the backward writes zeros for radii == 0, RAST/rasterize_points.cu:167-184 zero
             initialises them), so leaving those rows out of the reduction is exact.
    Cost model: one small all-reduce of the mask (4 P bytes), one host read of the row count (the ranks must agree on the
    packed size), a gather / scatter pass over the visible rows, and an all-reduce of K x D floats instead of P x D.  At
    BASELINE configs[2] about 60 % of the Gaussians are visible per view; the saving grows with the number of ranks (ring
    all-reduce is per-link bound on xGMI) and shrinks with the number of views per rank.  Returns K."""
    if not (dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1):
        return int(visible.sum().item())
    gs = [g for g in grads if g is not None and g.numel() > 0]
    P = int(visible.numel())
    for g in gs:
        if g.shape[0] != P or not g.is_contiguous():
            raise ValueError("sparse_all_reduce_grads takes contiguous per-Gaussian tensors [P, ...]")
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
