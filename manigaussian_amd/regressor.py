"""Gaussian-regressor epilogue: everything between the regressor's last Linear and the rasterizer.

Reference: agents/manigaussian_bc/models_embed.py:233-253 -- split the 26-vector into (xyz 3, opacity 1, scale 3, rot 4,
f_dc 3, feature 3, f_rest 9), scale = clamp_max(exp(.), 0.05), opacity = sigmoid, rot = F.normalize, sh = cat(f_dc, f_rest)
[N,4,3], xyz = xyz_in + delta -- and agents/manigaussian_bc/gaussian_renderer/__init__.py:66-68 -- the language feature is
L2-normalised with a 1e-12 guard right before rasterization.  ~10 torch kernels + 3 cats there; one HIP pass each way here.
"""
import ctypes

import torch

from . import _lib

RAW_DIM = 26


def _stream(dev):
    return ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)


class _Epilogue(torch.autograd.Function):
    @staticmethod
    def forward(ctx, raw, xyz_in):
        L = _lib.lib()
        if not raw.is_cuda:
            raise RuntimeError("gaussian_epilogue needs tensors on a HIP device; there is no CPU path")
        if raw.size(-1) != RAW_DIM:
            raise RuntimeError(f"expected a [..., {RAW_DIM}] regressor output, got {tuple(raw.shape)}")
        lead = raw.shape[:-1]
        dev = raw.device
        r = raw.float().contiguous().reshape(-1, RAW_DIM)
        x = xyz_in.float().contiguous().reshape(-1, 3)
        N = r.size(0)
        o = dict(dtype=torch.float32, device=dev)
        xyz, opacity, scale, rot = torch.empty((N, 3), **o), torch.empty((N, 1), **o), torch.empty((N, 3), **o), torch.empty((N, 4), **o)
        sh, feat, feat_n = torch.empty((N, 4, 3), **o), torch.empty((N, 3), **o), torch.empty((N, 3), **o)
        with torch.cuda.device(dev):
            _lib.check(L.mgs_regress_epilogue_forward(N, r.data_ptr(), x.data_ptr(), xyz.data_ptr(), opacity.data_ptr(),
                                                      scale.data_ptr(), rot.data_ptr(), sh.data_ptr(), feat.data_ptr(),
                                                      feat_n.data_ptr(), _stream(dev)), "regress_epilogue_forward")
        ctx.save_for_backward(r)
        ctx.lead = lead
        ctx.set_materialize_grads(False)
        rs = lambda t, *tail: t.reshape(*lead, *tail)  # noqa: E731
        return rs(xyz, 3), rs(opacity, 1), rs(scale, 3), rs(rot, 4), rs(sh, 4, 3), rs(feat, 3), rs(feat_n, 3)

    @staticmethod
    def backward(ctx, g_xyz, g_opacity, g_scale, g_rot, g_sh, g_feat, g_feat_n):
        L = _lib.lib()
        (r,) = ctx.saved_tensors
        dev = r.device
        N = r.size(0)
        c = lambda t: None if t is None else t.float().contiguous()  # noqa: E731
        gs = [c(g_xyz), c(g_opacity), c(g_scale), c(g_rot), c(g_sh), c(g_feat), c(g_feat_n)]
        g_raw = torch.empty((N, RAW_DIM), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            _lib.check(L.mgs_regress_epilogue_backward(N, r.data_ptr(), *[None if t is None else t.data_ptr() for t in gs],
                                                       g_raw.data_ptr(), _stream(dev)), "regress_epilogue_backward")
        g_in = None if g_xyz is None else g_xyz  # xyz = xyz_in + delta
        return g_raw.reshape(*ctx.lead, RAW_DIM), g_in


def gaussian_epilogue(raw, xyz_in):
    """raw [..., 26] regressor output, xyz_in [..., 3] -> dict(xyz, opacity, scale, rot, sh [...,4,3], feature,
    feature_normalized); differentiable w.r.t. both inputs."""
    xyz, opacity, scale, rot, sh, feat, feat_n = _Epilogue.apply(raw, xyz_in)
    return dict(xyz=xyz, opacity=opacity, scale=scale, rot=rot, sh=sh, feature=feat, feature_normalized=feat_n)
