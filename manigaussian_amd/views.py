"""Multi-view batches: V views of ONE Gaussian set rasterized in one call.

The reference renders strictly one view per call (agents/manigaussian_bc/neural_rendering.py:386 `assert bs == 1`,
and again for the next-frame scene, :324); ManiGaussian's BASELINE configs render 4-16 views per step.  At 128x128 a
single view cannot fill 256 CUs (64 tiles), so batching is the natural unit on MI355X: the views are stacked into an
atlas, every launch covers all of them, and the per-Gaussian gradients are summed over the views on the device.

    rast = GaussianRasterizerBatch([settings_0, ..., settings_{V-1}])     # same H, W, sh_degree, scale_modifier, bg
    color, feature, radii = rast(means3D, means2D, opacities, shs=..., language_feature_precomp=..., scales=..., rotations=...)
    # color [V,3,H,W], feature [V,F,H,W] (or [1]), radii [V,P] int32; means2D: [V,P,3] gradient holder (or None)

Results per view are those of GaussianRasterizer (same kernels); gradients w.r.t. the Gaussian parameters are the sums
over the views; `means2D.grad` is per view.
"""
import ctypes
from typing import Sequence

import torch
import torch.nn as nn

from . import _C, _lib, _state
from .rasterizer import GaussianRasterizationSettings, _EMPTY

_F32 = torch.float32


def _layout(L, P, M, F, V, precomp):
    scratch_f = (L.mgs_views_backward_scratch_bytes(P, M, F, V) + 3) // 4
    ncol = P if precomp else V * P
    sizes = [scratch_f, 3 * ncol, F * P, 3 * P, 3 * V * P, P, 6 * P, 3 * M * P, 3 * P, 4 * P, 4]
    accum_bytes = ((scratch_f + 3 * ncol + F * P) * 4 + 15) // 16 * 16
    return sizes, accum_bytes


class _RasterizeViews(torch.autograd.Function):
    @staticmethod
    def forward(ctx, means3D, means2D, sh, colors_precomp, language_feature, opacities, scales, rotations, cov3D_precomp,
                settings):
        L = _lib.lib()
        s0 = settings[0]
        V = len(settings)
        if not means3D.is_cuda:
            raise RuntimeError("GaussianRasterizerBatch needs tensors on a HIP device; there is no CPU path")
        dev = means3D.device
        P, H, W = int(means3D.size(0)), int(s0.image_height), int(s0.image_width)
        f = lambda t, n: _C._f32c(t, n, dev)  # noqa: E731
        means3D, sh, colors_precomp = f(means3D, "means3D"), f(sh, "sh"), f(colors_precomp, "colors_precomp")
        opacities, scales, rotations = f(opacities, "opacities"), f(scales, "scales"), f(rotations, "rotations")
        cov3D_precomp, bg = f(cov3D_precomp, "cov3D_precomp"), f(s0.bg, "bg")
        M = int(sh.size(1)) if sh.numel() else 0
        inc = bool(s0.include_feature)
        F = F_user = 0
        if inc:
            language_feature = f(language_feature, "language_feature_precomp")
            F_user = int(language_feature.size(1))
            F = _C._padded_F(F_user)
            if F != F_user:
                language_feature = torch.nn.functional.pad(language_feature, (0, F - F_user))
            language_feature = _C._aligned16(language_feature)
        views = (_lib.MgsView * V)()
        keep = []
        for v, s in enumerate(settings):
            vm, pm, cp = f(s.viewmatrix, "viewmatrix"), f(s.projmatrix, "projmatrix"), f(s.campos, "campos")
            keep += [vm, pm, cp]
            views[v].tanfovx, views[v].tanfovy = float(s.tanfovx), float(s.tanfovy)
            views[v].viewmatrix, views[v].projmatrix, views[v].campos = vm.data_ptr(), pm.data_ptr(), cp.data_ptr()
        with _C._on_device(dev):
            u8 = dict(dtype=torch.uint8, device=dev)
            out_color = torch.empty((V, 3, H, W), dtype=_F32, device=dev)
            out_feat = torch.empty((V, F, H, W), dtype=_F32, device=dev) if inc else torch.zeros((1,), dtype=_F32, device=dev)
            radii = torch.empty((V, P), dtype=torch.int32, device=dev)
            if P == 0:
                out_color.zero_()
                out_feat.zero_()
            st = _state.device_state(dev)
            capturing = _C._capturing()
            if not capturing:
                st.drain()
            opts = dict(_lib.DEFAULT_OPTIONS)
            key = ("views", V, P, W, H, F, opts["tight_bins"])
            guess = st.guess(key)
            T1 = ((W + 15) // 16) * ((H + 15) // 16)
            cap_worst = V * P * T1  # every Gaussian in every tile of every view
            worst_bytes = L.mgs_views_binning_bytes2(cap_worst, 0, W, H, F, V) if 0 < cap_worst < (1 << 30) else 0
            # (the budget is charged against the worst cases live forwards of the device still hold: _state.hold)
            cannot_overflow = (worst_bytes > 0 and worst_bytes <= _state.safe_bytes(dev) and
                               (_state.forward_mode() == "async" or _state.worst_case_fits(st.index, worst_bytes, dev)))
            if cannot_overflow:
                guess = (cap_worst, L.mgs_views_chunk_pool_max(cap_worst, W, H, V))
            lazy = guess is not None and _state.lazy_allowed(cannot_overflow) and not s0.prefiltered
            if capturing and not lazy:
                raise RuntimeError("capturing a batched forward into a HIP graph needs the asynchronous path: "
                                   "manigaussian_amd.set_forward_mode('async'), then run this shape eagerly (twice) first so "
                                   "that its workspace sizes are known")
            if lazy:
                cap, pool = guess
            else:
                m = st.marks.get(key)
                cap, pool = (m[0] + m[0] // 4 + 4096 if m else 4 * V * P + 4096), 0
            geom = torch.empty((L.mgs_views_geom_bytes(P, M, W, H, V),), **u8)
            img = torch.empty((L.mgs_views_img_bytes(W, H, V),), **u8)
            want = any(ctx.needs_input_grad[:9]) and P > 0
            grad_buffer = None
            a = _lib.MgsRasterArgs()
            handle = 0
            binning = _EMPTY
            while P > 0:
                # (+ room for the preprocess to write the tile keys itself: no bin scatter launch, include/mgsplat.h)
                binning = torch.empty((((L.mgs_views_binning_bytes2(cap, pool, W, H, F, V) + 255) & ~255) +
                                       (0 if cap == cap_worst else L.mgs_binning_direct_extra(P, V, W, H)),), **u8)
                if lazy and cannot_overflow and want and cap == cap_worst:
                    _state.hold(st.index, worst_bytes, binning)
                _C._fill_args(a, P=P, D=int(s0.sh_degree), M=M, F=F, W=W, H=H, tanfovx=0.0, tanfovy=0.0,
                              scale_modifier=float(s0.scale_modifier), prefiltered=s0.prefiltered, debug=False,
                              include_feature=inc, background=bg, means3D=means3D, sh=sh, colors=colors_precomp,
                              language_feature=language_feature, opacity=opacities, scales=scales, rotations=rotations,
                              cov3D_precomp=cov3D_precomp, viewmatrix=None, projmatrix=None, campos=None, geom=geom,
                              binning=binning, img=img)
                _lib.fill_options(a, opts)
                m_ = st.marks.get(key)
                if m_ is not None:
                    _lib.auto_seg(a, opts, m_[0], V * T1)
                slot_ptr, tag = st.take_slot()
                a.binning_capacity, a.chunk_pool, a.status_tag, a.async_forward = cap, pool, tag, 1 if lazy else 0
                if want and grad_buffer is None:
                    sizes, accum_bytes = _layout(L, P, M, F, V, colors_precomp.numel() != 0)
                    grad_buffer = torch.empty((sum(sizes),), dtype=_F32, device=dev)
                if want:
                    a.bwd_accum, a.bwd_accum_bytes = grad_buffer.data_ptr(), accum_bytes
                rc, R = _C._launch_forward(L, a, (views, V), radii, out_color, out_feat if inc else None, slot_ptr,
                                           _C._stream(dev))
                if rc == _lib.MGS_NEED_CAPACITY:  # (waiting path) the guess was too small: run the batch again with room for R
                    st.learn(key, _C._binned_now(slot_ptr))  # (R is the reference's count; the marks hold binned counts)
                    cap, pool = R + R // 4 + 4096, 0
                    continue
                _lib.check(rc, "rasterize views")
                pending = _state.Pending(a, V, slot_ptr, key, captured=capturing, recoverable=want and not capturing)
                st.add(pending)
                handle = _C.ForwardHandle(a, opts, pending, R, (views, language_feature), views=(views, V),
                                          outs=(_C._weak(out_color), _C._weak(out_feat) if inc and F == F_user else None))
                break
        ctx.settings, ctx.num_rendered, ctx.dims = settings, handle, (P, M, F, F_user, V, H, W)
        ctx.grad_buffer = grad_buffer
        ctx.mark_non_differentiable(radii)
        ctx.set_materialize_grads(False)
        ctx.save_for_backward(colors_precomp, language_feature if inc else _EMPTY, means3D, scales, rotations,
                              cov3D_precomp, radii, sh, geom, binning, img, bg, *keep)
        if inc and F != F_user:
            out_feat = out_feat[:, :F_user].contiguous()
        return out_color, out_feat, radii

    @staticmethod
    def backward(ctx, g_color, g_feat, _g_radii):
        L = _lib.lib()
        (colors_precomp, language_feature, means3D, scales, rotations, cov3D_precomp, radii, sh, geom, binning, img, bg,
         *cams) = ctx.saved_tensors
        P, M, F, F_user, V, H, W = ctx.dims
        settings = ctx.settings
        s0 = settings[0]
        dev = means3D.device
        inc = bool(s0.include_feature)
        if P == 0 or (g_color is None and g_feat is None):
            return (None,) * 10
        if g_color is None:
            g_color = torch.zeros((V, 3, H, W), dtype=_F32, device=dev)
        g_color = _C._f32c(g_color, "dL_dout_color", dev)
        if inc:
            if g_feat is None:
                g_feat = torch.zeros((V, F_user, H, W), dtype=_F32, device=dev)
            g_feat = _C._f32c(g_feat, "dL_dout_language_feature", dev)
            if F != F_user:
                g_feat = torch.cat([g_feat, g_feat.new_zeros((V, F - F_user, H, W))], 1)
        precomp = colors_precomp.numel() != 0
        with _C._on_device(dev):
            sizes, _ = _layout(L, P, M, F, V, precomp)
            grad_buffer, ctx.grad_buffer = ctx.grad_buffer, None
            prezeroed = grad_buffer is not None and grad_buffer.numel() == sum(sizes)
            flat = grad_buffer if prezeroed else torch.empty((sum(sizes),), dtype=_F32, device=dev)
            (scratch, d_colors, d_feat, d_means3D, d_means2D, d_opacity, d_cov3D, d_sh, d_scales, d_rot,
             _pad) = flat.split_with_sizes(sizes)
            handle = ctx.num_rendered
            a, views = handle.a, handle.views[0]  # the forward's arguments (their tensors are saved in ctx / handle.keep)
            count = _C._settle(handle, radii, dev, handle.num_rendered_nowait())
            a.accum_prezeroed = 1 if prezeroed else 0
            _lib.check(L.mgs_rasterize_backward_views(
                ctypes.byref(a), V, views, count, radii.data_ptr(), g_color.data_ptr(),
                g_feat.data_ptr() if inc else None, d_means2D.data_ptr(), None, d_opacity.data_ptr(),
                d_colors.data_ptr(), d_feat.data_ptr() if inc else None, d_means3D.data_ptr(), d_cov3D.data_ptr(),
                _C._ptr(d_sh), d_scales.data_ptr(), d_rot.data_ptr(), scratch.data_ptr(), scratch.numel() * 4,
                _C._stream(dev)), "rasterize views (backward)")
        d_feat = d_feat.view(P, F) if inc else None
        if inc and F != F_user:
            d_feat = d_feat[:, :F_user].contiguous()
        d_colors = d_colors.view(P, 3) if precomp else None  # per-view colour gradients only feed the SH backward
        return (d_means3D.view(P, 3), d_means2D.view(V, P, 3), d_sh.view(P, M, 3) if M else None, d_colors, d_feat,
                d_opacity.view(P, 1), d_scales.view(P, 3) if scales.numel() else None,
                d_rot.view(P, 4) if rotations.numel() else None,
                d_cov3D.view(P, 6) if cov3D_precomp.numel() else None, None)


class GaussianRasterizerBatch(nn.Module):
    """V views per call; see the module docstring.  Argument names and exclusivity rules are GaussianRasterizer's
    (RAST/diff_gaussian_rasterization/__init__.py:197-233)."""

    def __init__(self, raster_settings: Sequence[GaussianRasterizationSettings]):
        super().__init__()
        rs = list(raster_settings)
        if not 1 <= len(rs) <= _lib.MAX_VIEWS:
            raise ValueError(f"GaussianRasterizerBatch takes 1..{_lib.MAX_VIEWS} views, got {len(rs)}")
        s0 = rs[0]
        for s in rs[1:]:
            same = (s.image_height == s0.image_height and s.image_width == s0.image_width and s.sh_degree == s0.sh_degree
                    and s.scale_modifier == s0.scale_modifier and s.include_feature == s0.include_feature
                    and s.prefiltered == s0.prefiltered and (s.bg is s0.bg or torch.equal(s.bg, s0.bg)))
            if not same:
                raise ValueError("all views of a batch must share image size, sh_degree, scale_modifier, bg, "
                                 "include_feature and prefiltered")
        self.raster_settings = tuple(rs)

    def forward(self, means3D, means2D, opacities, shs=None, colors_precomp=None, language_feature_precomp=None,
                scales=None, rotations=None, cov3D_precomp=None):
        if (shs is None) == (colors_precomp is None):
            raise Exception('Please provide excatly one of either SHs or precomputed colors!')
        has_sr = scales is not None and rotations is not None
        any_sr = scales is not None or rotations is not None
        if (not has_sr and cov3D_precomp is None) or (any_sr and cov3D_precomp is not None):
            raise Exception('Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!')
        e = lambda t: _EMPTY if t is None else t  # noqa: E731
        if means2D is None:
            means2D = _EMPTY
        return _RasterizeViews.apply(means3D, means2D, e(shs), e(colors_precomp), e(language_feature_precomp), opacities,
                                     e(scales), e(rotations), e(cov3D_precomp), self.raster_settings)
