"""Public rasterizer API: the drop-in for the reference's `diff_gaussian_rasterization` Python module
(RAST/diff_gaussian_rasterization/__init__.py, RAST = third_party/gaussian-splatting/submodules/
diff-gaussian-rasterization).

Same surface, field order, positional/keyword order, return tuple, gradient order and exception
messages:
  GaussianRasterizationSettings  (13 fields, __init__.py:166-179)
  GaussianRasterizer(raster_settings).forward(means3D, means2D, opacities, shs=None, colors_precomp=None,
        language_feature_precomp=None, scales=None, rotations=None, cov3D_precomp=None)
        -> (color [3,H,W], language_feature [F,H,W] or [1], radii [P] int32)      (__init__.py:197-233)
  GaussianRasterizer.markVisible(positions) -> bool [P]                             (__init__.py:186-195)
  rasterize_gaussians(...)                                                          (__init__.py:21-44)

The native side is libmgsplat.so (hand-written HIP for gfx950) reached through manigaussian_amd._C.
"""
from typing import NamedTuple

import torch
import torch.nn as nn

from . import _C


class GaussianRasterizationSettings(NamedTuple):
    image_height: int
    image_width: int
    tanfovx: float
    tanfovy: float
    bg: torch.Tensor
    scale_modifier: float
    viewmatrix: torch.Tensor
    projmatrix: torch.Tensor
    sh_degree: int
    campos: torch.Tensor
    prefiltered: bool
    debug: bool
    include_feature: bool


def _snapshot(args):
    """CPU copy of the native-call arguments, taken before the call so a crash cannot corrupt it
    (reference: cpu_deep_copy_tuple, __init__.py:17-19)."""
    return tuple(x.detach().cpu().clone() if isinstance(x, torch.Tensor) else x for x in args)


def _call_native(fn, args, debug, dump_name, what):
    if not debug:
        return fn(*args)
    saved = _snapshot(args)
    try:
        return fn(*args)
    except Exception:
        torch.save(saved, dump_name)
        print(f"\nAn error occured in {what}. Please forward {dump_name} for debugging.")
        raise


class _RasterizeGaussians(torch.autograd.Function):
    """autograd glue (reference: __init__.py:46-164).  Saves the same eleven tensors; `opacities` is not
    saved -- the backward reads it from the geometry workspace like the reference does."""

    @staticmethod
    def forward(ctx, means3D, means2D, sh, colors_precomp, language_feature_precomp, opacities, scales, rotations,
                cov3Ds_precomp, raster_settings):
        s = raster_settings
        native_args = (s.bg, means3D, colors_precomp, language_feature_precomp, opacities, scales, rotations,
                       s.scale_modifier, cov3Ds_precomp, s.viewmatrix, s.projmatrix, s.tanfovx, s.tanfovy,
                       s.image_height, s.image_width, sh, s.sh_degree, s.campos, s.prefiltered, s.debug,
                       s.include_feature)
        # when a backward will follow, the forward also allocates its gradient buffer and zeroes the accumulator
        # block inside the preprocess kernel (saves the backward a 17 MB fill on the critical path at C3)
        want = any(ctx.needs_input_grad[:9])
        (num_rendered, color, language_feature, radii, geomBuffer, binningBuffer, imgBuffer, grad_buffer) = _call_native(
            _C._forward, native_args + (want,), s.debug, "snapshot_fw.dump", "forward")
        ctx.grad_buffer = grad_buffer
        ctx.raster_settings = s
        ctx.num_rendered = num_rendered
        ctx.mark_non_differentiable(radii)     # int32: no gradient, and no zeros_like(radii) fill per backward
        ctx.set_materialize_grads(False)       # an unused output arrives as None instead of a zero tensor
        ctx.save_for_backward(colors_precomp, language_feature_precomp, means3D, scales, rotations, cov3Ds_precomp,
                              radii, sh, geomBuffer, binningBuffer, imgBuffer)
        return color, language_feature, radii

    @staticmethod
    def backward(ctx, grad_out_color, grad_out_language_feature, _grad_radii):
        s = ctx.raster_settings
        (colors_precomp, language_feature_precomp, means3D, scales, rotations, cov3Ds_precomp, radii, sh, geomBuffer,
         binningBuffer, imgBuffer) = ctx.saved_tensors
        if grad_out_color is None and grad_out_language_feature is None:
            return (None,) * 10
        if grad_out_color is None:
            grad_out_color = torch.zeros((3, s.image_height, s.image_width), dtype=torch.float32, device=means3D.device)
        if grad_out_language_feature is None:
            grad_out_language_feature = torch.zeros((language_feature_precomp.size(1) if s.include_feature else 1,
                                                     s.image_height, s.image_width) if s.include_feature else (1,),
                                                    dtype=torch.float32, device=means3D.device)
        native_args = (s.bg, means3D, radii, colors_precomp, language_feature_precomp, scales, rotations,
                       s.scale_modifier, cov3Ds_precomp, s.viewmatrix, s.projmatrix, s.tanfovx, s.tanfovy,
                       grad_out_color, grad_out_language_feature, sh, s.sh_degree, s.campos, geomBuffer,
                       ctx.num_rendered, binningBuffer, imgBuffer, s.debug, s.include_feature)
        grad_buffer, ctx.grad_buffer = ctx.grad_buffer, None  # pre-zeroed for ONE backward; a second one allocates + fills
        (g_means2D, g_colors, g_feature, g_opacities, g_means3D, g_cov3D, g_sh, g_scales, g_rotations) = _call_native(
            _C._backward, native_args + (grad_buffer,), s.debug, "snapshot_bw.dump", "backward")
        # order of forward's inputs (reference: __init__.py:151-162)
        return (g_means3D, g_means2D, g_sh, g_colors, g_feature, g_opacities, g_scales, g_rotations, g_cov3D, None)


def rasterize_gaussians(means3D, means2D, sh, colors_precomp, language_feature_precomp, opacities, scales, rotations,
                        cov3Ds_precomp, raster_settings):
    # The compiled binding first (csrc/mgs_torch.cpp: the reference's own binding is compiled too,
    # RAST/rasterize_points.cu:35-225): marshalling, allocation and the autograd node in C++ over the same C ABI.  It returns
    # None for calls it does not handle (debug / prefiltered, HIP-graph capture, padded feature widths, non-contiguous or
    # non-fp32 inputs, the first calls of a shape in "async" mode, "blocking" mode): the ctypes shim below does those.
    ext = _C.compiled()
    if ext is not None:
        s = raster_settings
        try:
            out = ext.rasterize(means3D, means2D, sh, colors_precomp, language_feature_precomp, opacities, scales, rotations,
                                cov3Ds_precomp, s.bg, s.viewmatrix, s.projmatrix, s.campos, s.image_height, s.image_width,
                                s.tanfovx, s.tanfovy, s.scale_modifier, s.sh_degree, s.prefiltered, s.debug, s.include_feature)
        except TypeError:  # an argument the binding's signature does not convert (None for a tensor, ...): the shim's rules apply
            out = None
        if out is not None:
            return out
    return _RasterizeGaussians.apply(means3D, means2D, sh, colors_precomp, language_feature_precomp, opacities,
                                     scales, rotations, cov3Ds_precomp, raster_settings)


_EMPTY = torch.Tensor([])  # the reference passes torch.Tensor([]) for absent inputs (__init__.py:207-219)


def _or_empty(t):
    return _EMPTY if t is None else t


class GaussianRasterizer(nn.Module):
    def __init__(self, raster_settings):
        super().__init__()
        self.raster_settings = raster_settings

    def markVisible(self, positions):
        with torch.no_grad():
            s = self.raster_settings
            return _C.mark_visible(positions, s.viewmatrix, s.projmatrix)

    def forward(self, means3D, means2D, opacities, shs=None, colors_precomp=None, language_feature_precomp=None,
                scales=None, rotations=None, cov3D_precomp=None):
        # messages kept verbatim (typo included) from the reference, __init__.py:202,205
        if (shs is None) == (colors_precomp is None):
            raise Exception('Please provide excatly one of either SHs or precomputed colors!')
        has_sr = scales is not None and rotations is not None
        any_sr = scales is not None or rotations is not None
        if (not has_sr and cov3D_precomp is None) or (any_sr and cov3D_precomp is not None):
            raise Exception('Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!')
        return rasterize_gaussians(means3D, means2D, _or_empty(shs), _or_empty(colors_precomp),
                                   _or_empty(language_feature_precomp), opacities, _or_empty(scales),
                                   _or_empty(rotations), _or_empty(cov3D_precomp), self.raster_settings)
