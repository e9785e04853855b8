"""The three functions the reference's pybind module `diff_gaussian_rasterization._C` exports
(RAST/ext.cpp:15-19), with the same names, argument order and return tuples, implemented over the
C ABI of libmgsplat.so.  RAST = third_party/gaussian-splatting/submodules/diff-gaussian-rasterization.

torch is plumbing here: it owns device memory (outputs + the three opaque byte workspaces that the
reference also returns as uint8 tensors) and supplies the current HIP stream.
"""
import ctypes

import torch

from . import _lib

_F32 = torch.float32


def _padded_F(F: int) -> int:
    for s in _lib.SUPPORTED_F:
        if F <= s:
            return s
    raise RuntimeError(f"language feature width {F} > {_lib.SUPPORTED_F[-1]} is not supported")


def _ptr(t):
    return None if (t is None or t.numel() == 0) else t.data_ptr()


def _f32c(t, name, dev):
    """contiguous float32 on dev; empty tensors pass through (they become NULL pointers, like the
    reference's .data<float>() of an empty tensor)."""
    if t.dtype is _F32 and t.device == dev and t.is_contiguous():  # the hot path: nothing to do
        return t
    if t.numel() == 0:
        return t
    if t.dtype != _F32:
        raise RuntimeError(f"expected scalar type Float but found {t.dtype} for {name}")
    if t.device != dev:
        raise RuntimeError(f"{name} is on {t.device} but means3D is on {dev}")
    return t.contiguous()


def _stream(dev):
    return ctypes.c_void_p(torch._C._cuda_getCurrentRawStream(dev.index if dev.index is not None else torch.cuda.current_device()))


class _on_device:
    """`with torch.cuda.device(dev)` only when dev is not already current (the context manager costs ~10 us)."""

    def __init__(self, dev):
        idx = dev.index if dev.index is not None else torch.cuda.current_device()
        self.ctx = None if idx == torch.cuda.current_device() else torch.cuda.device(idx)

    def __enter__(self):
        if self.ctx is not None:
            self.ctx.__enter__()

    def __exit__(self, *a):
        if self.ctx is not None:
            self.ctx.__exit__(*a)


def _fill_args(a, *, P, D, M, F, W, H, tanfovx, tanfovy, scale_modifier, prefiltered, debug, include_feature,
               background, means3D, sh, colors, language_feature, opacity, scales, rotations, cov3D_precomp,
               viewmatrix, projmatrix, campos, geom, binning, img):
    a.P, a.D, a.M, a.F, a.W, a.H = P, D, M, F, W, H
    a.tanfovx, a.tanfovy, a.scale_modifier = tanfovx, tanfovy, scale_modifier
    a.prefiltered, a.debug, a.include_feature = int(bool(prefiltered)), int(bool(debug)), int(bool(include_feature))
    a.background, a.means3D, a.shs, a.colors_precomp = _ptr(background), _ptr(means3D), _ptr(sh), _ptr(colors)
    a.language_feature = _ptr(language_feature) if include_feature else None
    a.opacities, a.scales, a.rotations, a.cov3D_precomp = _ptr(opacity), _ptr(scales), _ptr(rotations), _ptr(cov3D_precomp)
    a.viewmatrix, a.projmatrix, a.campos = _ptr(viewmatrix), _ptr(projmatrix), _ptr(campos)
    a.geom, a.geom_bytes = _ptr(geom), 0 if geom is None else geom.numel()
    a.binning, a.binning_bytes = _ptr(binning), 0 if binning is None else binning.numel()
    a.img, a.img_bytes = _ptr(img), 0 if img is None else img.numel()


# Per-device host state of the sync-free forward: one pinned 8-byte status word (the device reports
# {flags, num_rendered} through it) and the high-water mark of num_rendered per problem shape, which sizes
# the binning workspace BEFORE the count is known (the reference sizes it after a blocking read-back).
_DEV_STATE = {}
_CAPACITY = {}  # device -> {shape key: capacity}; shared by all threads (a stale read only costs one retry)


def _dev_state(dev):
    """The status word belongs to ONE in-flight forward: it is per (device, calling thread) -- ctypes drops the
    GIL during the native call, so two Python threads can be inside mgs_rasterize_forward at once."""
    import threading
    key = (dev, threading.get_ident())
    st = _DEV_STATE.get(key)
    if st is None:
        pin = torch.zeros(2, dtype=torch.int64).pin_memory()
        st = {"status": pin, "status_ptr": pin.data_ptr(), "cap": _CAPACITY.setdefault(dev, {})}
        _DEV_STATE[key] = st
    return st


def _capacity_guess(st, key, P):
    cap = st["cap"].get(key)
    return cap if cap is not None else 4 * P + 4096


def _remember_capacity(st, key, R):
    want = R + R // 4 + 4096  # 25 % head-room over the largest count seen for this shape
    if st["cap"].get(key, 0) < want:
        st["cap"][key] = want


def _grad_layout(L, P, M, F):
    """Float offsets of the backward's single allocation: [scratch | dL_dcolors | dL_dfeature | means3D | means2D |
    opacity | cov3D | sh | scales | rotations | pad].  The first three regions are the accumulators."""
    scratch_f = (L.mgs_backward_scratch_bytes(P, M, F) + 3) // 4
    sizes = [scratch_f, 3 * P, F * P, 3 * P, 3 * P, P, 6 * P, 3 * M * P, 3 * P, 4 * P, 4]
    accum_bytes = ((scratch_f + 3 * P + F * P) * 4 + 15) // 16 * 16  # may reach into the next, fully rewritten, region
    return sizes, accum_bytes


def rasterize_gaussians(background, means3D, colors, language_feature, opacity, scales, rotations, scale_modifier,
                        cov3D_precomp, viewmatrix, projmatrix, tan_fovx, tan_fovy, image_height, image_width, sh,
                        degree, campos, prefiltered, debug, include_feature):
    """RasterizeGaussiansCUDA (RAST/rasterize_points.cu:35-128).
    Returns (num_rendered, out_color [3,H,W], out_language_feature [F,H,W] or [1], radii [P] int32,
             geomBuffer, binningBuffer, imgBuffer)."""
    return _forward(background, means3D, colors, language_feature, opacity, scales, rotations, scale_modifier,
                    cov3D_precomp, viewmatrix, projmatrix, tan_fovx, tan_fovy, image_height, image_width, sh, degree,
                    campos, prefiltered, debug, include_feature, False)[:7]


def _forward(background, means3D, colors, language_feature, opacity, scales, rotations, scale_modifier, cov3D_precomp,
             viewmatrix, projmatrix, tan_fovx, tan_fovy, image_height, image_width, sh, degree, campos, prefiltered,
             debug, include_feature, want_grad_buffer):
    """rasterize_gaussians + (want_grad_buffer) the backward's allocation, whose accumulator block the forward's
    preprocess kernel zeroes on the side: returns the 7-tuple + (grad_buffer or None,)."""
    L = _lib.lib()
    if means3D.ndimension() != 2 or means3D.size(1) != 3:
        raise RuntimeError("means3D must have dimensions (num_points, 3)")  # rasterize_points.cu:59-61
    if not means3D.is_cuda:
        raise RuntimeError("diff_gaussian_rasterization (MI355X build) needs tensors on a HIP device; "
                           "there is no CPU path")
    dev = means3D.device
    P, H, W = int(means3D.size(0)), int(image_height), int(image_width)
    means3D = _f32c(means3D, "means3D", dev)
    background = _f32c(background, "background", dev)
    colors = _f32c(colors, "colors_precomp", dev)
    opacity = _f32c(opacity, "opacities", dev)
    scales = _f32c(scales, "scales", dev)
    rotations = _f32c(rotations, "rotations", dev)
    cov3D_precomp = _f32c(cov3D_precomp, "cov3D_precomp", dev)
    viewmatrix = _f32c(viewmatrix, "viewmatrix", dev)
    projmatrix = _f32c(projmatrix, "projmatrix", dev)
    campos = _f32c(campos, "campos", dev)
    sh = _f32c(sh, "sh", dev)
    M = int(sh.size(1)) if sh.numel() != 0 else 0
    include_feature = bool(include_feature)
    F = F_user = 0
    if include_feature:
        if language_feature.ndimension() != 2 or language_feature.size(0) != P:
            raise RuntimeError("language_feature_precomp must have dimensions (num_points, F)")
        language_feature = _f32c(language_feature, "language_feature_precomp", dev)
        F_user = int(language_feature.size(1))
        F = _padded_F(F_user)
        if F != F_user:  # feature widths that are not compiled in: zero channels change nothing
            language_feature = torch.nn.functional.pad(language_feature, (0, F - F_user))

    with _on_device(dev):
        u8 = dict(dtype=torch.uint8, device=dev)
        if P == 0:  # rasterize_points.cu:92: empty workspaces, zero images
            out_color = torch.zeros((3, H, W), dtype=_F32, device=dev)
            out_feat = torch.zeros((F_user, H, W) if include_feature else (1,), dtype=_F32, device=dev)
            e = torch.empty((0,), **u8)
            return (0, out_color, out_feat, torch.zeros((0,), dtype=torch.int32, device=dev), e, e.clone(), e.clone(),
                    None)
        out_color = torch.empty((3, H, W), dtype=_F32, device=dev)
        out_feat = torch.empty((F, H, W), dtype=_F32, device=dev) if include_feature else \
            torch.zeros((1,), dtype=_F32, device=dev)
        radii = torch.empty((P,), dtype=torch.int32, device=dev)  # written for every Gaussian by the preprocess
        st = _dev_state(dev)
        key = (P, W, H, F)
        cap = _capacity_guess(st, key, P)
        geom = torch.empty((L.mgs_geom_bytes(P, M, W, H),), **u8)
        img = torch.empty((L.mgs_img_bytes(W, H),), **u8)
        binning = torch.empty((L.mgs_binning_bytes(cap, W, H, F),), **u8)
        a = _lib.MgsRasterArgs()
        _fill_args(a, P=P, D=int(degree), M=M, F=F, W=W, H=H, tanfovx=float(tan_fovx), tanfovy=float(tan_fovy),
                   scale_modifier=float(scale_modifier), prefiltered=prefiltered, debug=debug,
                   include_feature=include_feature, background=background, means3D=means3D, sh=sh, colors=colors,
                   language_feature=language_feature, opacity=opacity, scales=scales, rotations=rotations,
                   cov3D_precomp=cov3D_precomp, viewmatrix=viewmatrix, projmatrix=projmatrix, campos=campos,
                   geom=geom, binning=binning, img=img)
        grad_buffer = None
        if want_grad_buffer:
            sizes, accum_bytes = _grad_layout(L, P, M, F)
            grad_buffer = torch.empty((sum(sizes),), dtype=_F32, device=dev)
            a.bwd_accum, a.bwd_accum_bytes = grad_buffer.data_ptr(), accum_bytes
        stream = _stream(dev)
        nr = ctypes.c_int32(0)
        feat_ptr = out_feat.data_ptr() if include_feature else None
        rc = L.mgs_rasterize_forward(ctypes.byref(a), radii.data_ptr(), out_color.data_ptr(), feat_ptr,
                                     ctypes.byref(nr), st["status_ptr"], stream)
        R = int(nr.value)
        if rc == _lib.MGS_NEED_CAPACITY:  # first call for this shape, or the scene grew: bin + render again
            binning = torch.empty((L.mgs_binning_bytes(R + R // 4 + 4096, W, H, F),), **u8)
            a.binning, a.binning_bytes = binning.data_ptr(), binning.numel()
            rc = L.mgs_rasterize_forward_render(ctypes.byref(a), R, radii.data_ptr(), out_color.data_ptr(), feat_ptr,
                                                stream)
        _lib.check(rc, "rasterize_gaussians")
        _remember_capacity(st, key, R)
    if include_feature and F != F_user:
        out_feat = out_feat[:F_user].contiguous()
    return R, out_color, out_feat, radii, geom, binning, img, grad_buffer


def rasterize_gaussians_backward(background, means3D, radii, colors, language_feature, scales, rotations,
                                 scale_modifier, cov3D_precomp, viewmatrix, projmatrix, tan_fovx, tan_fovy,
                                 dL_dout_color, dL_dout_language_feature, sh, degree, campos, geomBuffer, R,
                                 binningBuffer, imageBuffer, debug, include_feature):
    """RasterizeGaussiansBackwardCUDA (RAST/rasterize_points.cu:130-225).
    Returns (dL_dmeans2D [P,3], dL_dcolors [P,3], dL_dlanguage_feature [P,F] or [1], dL_dopacity [P,1],
             dL_dmeans3D [P,3], dL_dcov3D [P,6], dL_dsh [P,M,3], dL_dscales [P,3], dL_drotations [P,4])."""
    return _backward(background, means3D, radii, colors, language_feature, scales, rotations, scale_modifier,
                     cov3D_precomp, viewmatrix, projmatrix, tan_fovx, tan_fovy, dL_dout_color,
                     dL_dout_language_feature, sh, degree, campos, geomBuffer, R, binningBuffer, imageBuffer, debug,
                     include_feature, None)


def _backward(background, means3D, radii, colors, language_feature, scales, rotations, scale_modifier, cov3D_precomp,
              viewmatrix, projmatrix, tan_fovx, tan_fovy, dL_dout_color, dL_dout_language_feature, sh, degree, campos,
              geomBuffer, R, binningBuffer, imageBuffer, debug, include_feature, grad_buffer):
    """rasterize_gaussians_backward; grad_buffer = the allocation _forward() handed out (accumulators already
    zeroed by the forward's preprocess kernel) or None."""
    L = _lib.lib()
    dev = means3D.device
    P = int(means3D.size(0))
    H, W = int(dL_dout_color.size(1)), int(dL_dout_color.size(2))
    M = int(sh.size(1)) if sh.numel() != 0 else 0
    include_feature = bool(include_feature)
    means3D = _f32c(means3D, "means3D", dev)
    colors = _f32c(colors, "colors_precomp", dev)
    scales = _f32c(scales, "scales", dev)
    rotations = _f32c(rotations, "rotations", dev)
    cov3D_precomp = _f32c(cov3D_precomp, "cov3D_precomp", dev)
    sh = _f32c(sh, "sh", dev)
    dL_dout_color = _f32c(dL_dout_color, "dL_dout_color", dev)
    F = F_user = 0
    if include_feature:
        language_feature = _f32c(language_feature, "language_feature_precomp", dev)
        F_user = int(language_feature.size(1))
        F = _padded_F(F_user)
        dL_dout_language_feature = _f32c(dL_dout_language_feature, "dL_dout_language_feature", dev)
        if F != F_user:
            language_feature = torch.nn.functional.pad(language_feature, (0, F - F_user))
            dL_dout_language_feature = torch.cat(
                [dL_dout_language_feature, dL_dout_language_feature.new_zeros((F - F_user, H, W))], 0)
    with _on_device(dev):
        if P == 0:
            z = lambda *s_: torch.zeros(s_, dtype=_F32, device=dev)  # noqa: E731
            return (z(0, 3), z(0, 3), z(0, F_user) if include_feature else z(1), z(0, 1), z(0, 3), z(0, 6),
                    z(0, M, 3), z(0, 3), z(0, 4))
        # ONE allocation for the scratch accumulators and every gradient: the three regions the render backward
        # accumulates into (acc8 | dL_dcolors | dL_dfeature) come first and are contiguous, so the library zeroes
        # them with a single fill; everything else is fully written by the kernels.
        sizes, _ = _grad_layout(L, P, M, F)
        prezeroed = grad_buffer is not None and grad_buffer.numel() == sum(sizes)
        flat = grad_buffer if prezeroed else torch.empty((sum(sizes),), dtype=_F32, device=dev)
        (scratch, g_colors, g_feat, g_means3D, g_means2D, g_opacity, g_cov3D, g_sh, g_scales, g_rot,
         _pad) = flat.split_with_sizes(sizes)
        a = _lib.MgsRasterArgs()
        a.accum_prezeroed = 1 if prezeroed else 0
        _fill_args(a, P=P, D=int(degree), M=M, F=F, W=W, H=H, tanfovx=float(tan_fovx), tanfovy=float(tan_fovy),
                   scale_modifier=float(scale_modifier), prefiltered=False, debug=debug,
                   include_feature=include_feature, background=_f32c(background, "background", dev),
                   means3D=means3D, sh=sh, colors=colors, language_feature=language_feature, opacity=None,
                   scales=scales, rotations=rotations, cov3D_precomp=cov3D_precomp,
                   viewmatrix=_f32c(viewmatrix, "viewmatrix", dev), projmatrix=_f32c(projmatrix, "projmatrix", dev),
                   campos=_f32c(campos, "campos", dev), geom=geomBuffer, binning=binningBuffer, img=imageBuffer)
        _lib.check(L.mgs_rasterize_backward(
            ctypes.byref(a), int(R), radii.data_ptr(), dL_dout_color.data_ptr(),
            _ptr(dL_dout_language_feature) if include_feature else None, g_means2D.data_ptr(), None,
            g_opacity.data_ptr(), g_colors.data_ptr(), _ptr(g_feat) if include_feature else None,
            g_means3D.data_ptr(), g_cov3D.data_ptr(), _ptr(g_sh), g_scales.data_ptr(), g_rot.data_ptr(),
            scratch.data_ptr(), scratch.numel() * 4, _stream(dev)), "rasterize_gaussians_backward")
        g_feat = g_feat.view(P, F) if include_feature else torch.zeros((1,), dtype=_F32, device=dev)
    if include_feature and F != F_user:
        g_feat = g_feat[:, :F_user].contiguous()
    return (g_means2D.view(P, 3), g_colors.view(P, 3), g_feat, g_opacity.view(P, 1), g_means3D.view(P, 3),
            g_cov3D.view(P, 6), g_sh.view(P, M, 3), g_scales.view(P, 3), g_rot.view(P, 4))


def mark_visible(means3D, viewmatrix, projmatrix):
    """markVisible (RAST/rasterize_points.cu:227-246): bool [P], True where view-space z > 0.2."""
    L = _lib.lib()
    if not means3D.is_cuda:
        raise RuntimeError("mark_visible needs tensors on a HIP device; there is no CPU path")
    dev = means3D.device
    P = int(means3D.size(0))
    present = torch.zeros((P,), dtype=torch.bool, device=dev)
    if P != 0:
        with torch.cuda.device(dev):
            m = _f32c(means3D, "means3D", dev)
            v = _f32c(viewmatrix, "viewmatrix", dev)
            p = _f32c(projmatrix, "projmatrix", dev)
            _lib.check(L.mgs_mark_visible(P, m.data_ptr(), v.data_ptr(), p.data_ptr(), present.data_ptr(),
                                          _stream(dev)), "mark_visible")
    return present
