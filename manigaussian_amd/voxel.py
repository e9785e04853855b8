"""Per-point latent: voxel-feature trilinear gather + NeRF positional encoding, fused.

Reference: agents/manigaussian_bc/models_embed.py:147-215 (world_to_canonical, sample_in_canonical_voxel = F.grid_sample with
align_corners=True, the concat with the positional code) and agents/manigaussian_bc/utils.py:133-169 (PositionalEncoding,
num_freqs 6, freq_factor pi, include_input).  `point_latent_pe(dec_fts, xyz, bounds)` returns latent [N, C + 39] =
[point_latent | z_feature]; gradient flows into dec_fts (the voxel features), the points are data.
"""
import ctypes

import torch

from . import _lib


def _stream(dev):
    return ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)


class _PointLatent(torch.autograd.Function):
    @staticmethod
    def forward(ctx, voxel, xyz, bounds, num_freqs, freq_factor):
        L = _lib.lib()
        if not voxel.is_cuda:
            raise RuntimeError("point_latent_pe needs tensors on a HIP device; there is no CPU path")
        if voxel.dim() != 5 or voxel.size(0) != 1:
            raise RuntimeError(f"expected voxel features [1, C, D, H, W], got {tuple(voxel.shape)}")
        dev = voxel.device
        vox = voxel.float().contiguous()
        pts = xyz.float().contiguous().reshape(-1, 3)
        _, C, D, H, W = vox.shape
        N = pts.size(0)
        out = torch.empty((N, C + 3 + 6 * num_freqs), dtype=torch.float32, device=dev)
        b = (ctypes.c_float * 6)(*[float(x) for x in bounds])
        with torch.cuda.device(dev):
            _lib.check(L.mgs_voxel_sample_pe_forward(N, C, D, H, W, int(num_freqs), float(freq_factor), b, vox.data_ptr(),
                                                     pts.data_ptr(), out.data_ptr(), _stream(dev)), "voxel_sample_pe_forward")
        ctx.save_for_backward(pts)
        ctx.meta = (C, D, H, W, tuple(float(x) for x in bounds), voxel.shape)
        return out

    @staticmethod
    def backward(ctx, g_out):
        L = _lib.lib()
        (pts,) = ctx.saved_tensors
        C, D, H, W, bounds, vshape = ctx.meta
        dev = pts.device
        g = g_out.float().contiguous()
        g_vox = torch.zeros(vshape, dtype=torch.float32, device=dev)
        b = (ctypes.c_float * 6)(*bounds)
        with torch.cuda.device(dev):
            _lib.check(L.mgs_voxel_sample_backward(pts.size(0), C, D, H, W, b, pts.data_ptr(), g.data_ptr(), g.size(1),
                                                   g_vox.data_ptr(), _stream(dev)), "voxel_sample_backward")
        return g_vox, None, None, None, None


def point_latent_pe(voxel_feat, xyz, coordinate_bounds, num_freqs=6, freq_factor=3.141592653589793):
    """voxel_feat [1,C,D,H,W], xyz [..., 3] world coordinates, coordinate_bounds (xmin,ymin,zmin,xmax,ymax,zmax)
    -> [N, C + 3 + 6*num_freqs] = cat(point_latent, PositionalEncoding(canon_xyz))  (models_embed.py:201-215)."""
    return _PointLatent.apply(voxel_feat, xyz, coordinate_bounds, num_freqs, freq_factor)
