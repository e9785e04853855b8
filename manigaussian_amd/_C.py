"""The three functions the reference's pybind module `diff_gaussian_rasterization._C` exports
(RAST/ext.cpp:15-19), with the same names, argument order and return tuples, implemented over the
C ABI of libmgsplat.so.  RAST = third_party/gaussian-splatting/submodules/diff-gaussian-rasterization.

torch is plumbing here: it owns device memory (outputs + the three opaque byte workspaces that the
reference also returns as uint8 tensors) and supplies the current HIP stream.
"""
import ctypes
import weakref

import torch

from . import _lib, _state

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


def _aligned16(t):
    """The render backward reads feature rows with 16-byte loads: a contiguous view that starts at an odd offset of its
    storage (a rare slice) is copied once."""
    return t if (t.numel() == 0 or t.data_ptr() % 16 == 0) else t.clone()


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


# ---- host-side caches: everything that is a pure function of the problem shape is computed once --------------------------------
_SIZES = {}      # (P, M, W, H) -> (geom bytes, img bytes), both rounded up to 256
_BIN_BYTES = {}  # (cap, pool, W, H, F) -> binning bytes
_WORST = {}      # (P, W, H, F, safe_bytes) -> (cannot_overflow, cap_worst, pool_worst)
_LAYOUTS = {}    # (P, M, F) -> (sizes, accum_bytes)
_TEMPLATES = {}  # shape + settings scalars + options version -> bytes of a pre-filled MgsRasterArgs
_EMPTY_U8 = {}   # device -> an empty uint8 tensor (placeholder for workspaces that live in another tensor's arena)
_SPLIT_WORKSPACES = False  # testing: geom / img / binning as three allocations (guard bands behind each, tests/test_gpu_parity.py)


def _up256(n):
    return (n + 255) & ~255


def _shape_sizes(L, P, M, W, H):
    k = (P, M, W, H)
    v = _SIZES.get(k)
    if v is None:
        v = _SIZES[k] = (_up256(L.mgs_geom_bytes(P, M, W, H)), _up256(L.mgs_img_bytes(W, H)))
    return v


def _bin_bytes(L, cap, pool, W, H, F, P=0):
    """Bytes of the binning workspace; P > 0: plus the room in which the forward preprocess writes the tile keys itself (no bin
    scatter launch; include/mgsplat.h mgs_binning_direct_extra) where the library offers it for the shape."""
    k = (cap, pool, W, H, F, P)
    v = _BIN_BYTES.get(k)
    if v is None:
        if len(_BIN_BYTES) > 4096:
            _BIN_BYTES.clear()
        v = L.mgs_binning_bytes2(cap, pool, W, H, F)
        if P > 0:
            extra = L.mgs_binning_direct_extra(P, 1, W, H)
            v = _up256(v) + extra if extra else v
        _BIN_BYTES[k] = v
    return v


def _worst_case(L, P, W, H, F, T, dev=None):
    budget = _state.safe_bytes(dev)
    k = (P, W, H, F, budget)
    v = _WORST.get(k)
    if v is None:
        cap_worst = P * T  # every Gaussian in every tile
        ok = cap_worst < (1 << 30) and L.mgs_binning_bytes2(cap_worst, 0, W, H, F) <= budget
        v = _WORST[k] = (ok, cap_worst, L.mgs_chunk_pool_max(cap_worst, W, H) if ok else 0)
    return v


def _template(P, D, M, F, W, H, tanfovx, tanfovy, scale_modifier, prefiltered, debug, include_feature):
    """A MgsRasterArgs with every field that does not change from call to call already filled (shape, camera scalars,
    options); _forward copies it and sets the pointers."""
    k = (P, D, M, F, W, H, tanfovx, tanfovy, scale_modifier, prefiltered, debug, include_feature, _lib.OPTIONS_VERSION[0])
    t = _TEMPLATES.get(k)
    if t is None:
        if len(_TEMPLATES) > 512:
            _TEMPLATES.clear()
        a = _lib.MgsRasterArgs()
        a.P, a.D, a.M, a.F, a.W, a.H = P, D, M, F, W, H
        a.tanfovx, a.tanfovy, a.scale_modifier = tanfovx, tanfovy, scale_modifier
        a.prefiltered, a.debug, a.include_feature = int(bool(prefiltered)), int(bool(debug)), int(bool(include_feature))
        opts = _lib.fill_options(a)
        t = _TEMPLATES[k] = (bytes(a), opts)
    return t


class ForwardHandle:
    """What a forward leaves for its backward: the filled MgsRasterArgs (reused, not rebuilt), the per-call options and
    the pending device report.  int(handle) blocks until the instance count is known (the reference returns it as an int):
    THE REFERENCE'S integer -- the (Gaussian, tile) instances of the 3-sigma rects, rasterizer_impl.cu:280-284 -- on every
    path (round 4: the asynchronous path handed out the count of instances actually binned); binned() is that other count."""
    __slots__ = ("a", "opts", "pending", "R", "keep", "views", "outs")

    def __init__(self, a, opts, pending, R, keep=None, views=None, outs=None):
        self.a, self.opts, self.pending, self.R = a, opts, pending, R
        self.keep = keep    # the tensors whose addresses `a` holds (converted / padded copies would otherwise be freed)
        self.views = views  # (MgsView array, V) of a batched forward, else None
        self.outs = outs    # weak references to (out_color, out_feature): a recovery re-renders into them if they still live

    def num_rendered_nowait(self) -> int:
        """The reference's count if the device has reported it, else -1 (never blocks)."""
        if self.R < 0 and self.pending is not None:
            self.pending.poll()
            self.R = self.pending.ref_rendered
        return self.R

    def _wait_report(self):
        p = self.pending
        spins = 0
        while p.num_rendered < 0 and p.poll() == _lib.MGS_PENDING:
            spins += 1
            if spins == 100000:  # ~0.1 s of polling: let the stream finish; a count that is still missing never comes
                torch.cuda.synchronize()
                if p.num_rendered < 0 and p.poll() == _lib.MGS_PENDING and p.num_rendered < 0:
                    raise RuntimeError("the rasterizer forward finished without reporting its instance count")

    def __int__(self):
        if self.R < 0 and self.pending is not None:
            self._wait_report()
            self.R = self.pending.ref_rendered
        return self.R

    __index__ = __int__

    def binned(self) -> int:
        """Instances actually binned (<= int(self) under tight_bins): what sizes a workspace and what the kernels move.
        Blocks like int()."""
        if self.pending is None:
            return max(int(self.R), 0)
        self._wait_report()
        return self.pending.num_rendered


def _capturing() -> bool:
    return torch.cuda.is_current_stream_capturing()


def _binned_now(slot_ptr):
    """The instance count in word 0 of a status slot whose report has arrived."""
    words, i = _state._WORDS[slot_ptr]
    return int(words[i]) & 0xffffffff


def _launch_forward(L, a, views, radii, out_color, out_feat, slot_ptr, stream):
    """mgs_rasterize_forward / mgs_rasterize_forward_views -> (rc, the reference's num_rendered or -1)."""
    nr = ctypes.c_int32(0)
    feat_ptr = out_feat.data_ptr() if (out_feat is not None and a.include_feature) else None
    if views is not None:
        rc = L.mgs_rasterize_forward_views(ctypes.byref(a), views[1], views[0], radii.data_ptr(), out_color.data_ptr(),
                                           feat_ptr, ctypes.byref(nr), slot_ptr, stream)
    else:
        rc = L.mgs_rasterize_forward(ctypes.byref(a), radii.data_ptr(), out_color.data_ptr(), feat_ptr, ctypes.byref(nr),
                                     slot_ptr, stream)
    return rc, int(nr.value)


def _weak(t):
    return None if t is None else weakref.ref(t)


def recover_forward(handle, radii, dev):
    """The asynchronous forward behind `handle` outgrew the workspace that was sized from earlier calls of its shape, and its
    report arrived before the backward was enqueued: bin and render it AGAIN on the blocking path (a workspace sized from the
    instance count the device reported, worst-case chunk pool: cannot overflow) into the same output tensors, so that the
    backward that follows runs on a complete forward state.  The reference can never be in this position
    (RAST/cuda_rasterizer/rasterizer_impl.cu:284-289 sizes the buffer from the count it waited for); this is the price of
    not waiting, paid only on the call where a scene grew past its head-room.  What cannot be repaired: whatever was computed
    from the incomplete images between that forward and this backward (the loss value, its cotangents) -- hence the warning
    _state issues when it reads the report."""
    L = _lib.lib()
    st = _state.device_state(dev)
    p, a = handle.pending, handle.a
    V = handle.views[1] if handle.views is not None else 0
    W, H, F = int(a.W), int(a.H), int(a.F) if a.include_feature else 0
    handshake_only = p.rc == _lib.MGS_RETRY_TABLE_INIT
    binning = None
    if not handshake_only:
        R = max(int(p.num_rendered), 0)
        cap = R + R // 4 + 4096
        nbytes = L.mgs_views_binning_bytes2(cap, 0, W, H, F, V) if V else L.mgs_binning_bytes2(cap, 0, W, H, F)
        binning = torch.empty((nbytes,), dtype=torch.uint8, device=dev)
    out_color = handle.outs[0]() if handle.outs and handle.outs[0] is not None else None
    out_feat = handle.outs[1]() if handle.outs and handle.outs[1] is not None else None
    lead = (V,) if V else ()
    if out_color is None:
        out_color = torch.empty(lead + (3, H, W), dtype=_F32, device=dev)
    if a.include_feature and out_feat is None:
        out_feat = torch.empty(lead + (F, H, W), dtype=_F32, device=dev)
    if handshake_only:
        # the scene did not outgrow anything: a workgroup of the preprocess launch gave up waiting for workgroup 0's zeroed
        # tables.  Same workspace, tables zeroed by a launch of their own this time (no workgroup waits for another).
        a.opt.set, a.opt.table_init, a.async_forward = 1, 1, 0
    else:
        a.binning, a.binning_bytes, a.binning_capacity, a.chunk_pool, a.async_forward = binning.data_ptr(), nbytes, cap, 0, 0
    a.bwd_accum, a.bwd_accum_bytes = None, 0  # the first run's preprocess zeroed the accumulators; nothing touched them since
    slot_ptr, tag = st.take_slot()
    a.status_tag = tag
    rc, R2 = _launch_forward(L, a, handle.views, radii, out_color, out_feat, slot_ptr, _stream(dev))
    _lib.check(rc, "rasterizer forward (re-run after a workspace overflow)")
    newp = _state.Pending(a, V, slot_ptr, p.key)
    st.add(newp)  # (the marks learn the binned count from its report)
    p.recovered = True
    handle.pending, handle.R = newp, R2
    if binning is not None:
        handle.keep = (handle.keep, binning)
    return R2


def _grad_layout(L, P, M, F):
    """Float offsets of the backward's single allocation: [scratch | dL_dcolors | dL_dfeature | means3D | opacity | sh |
    scales | rotations | cov3D | means2D | pad].  The first three regions are the accumulators; the gradients of the
    Gaussian PARAMETERS (colours or SH, features, means, opacity, scales, rotations) are contiguous, so one all-reduce
    over the span they cover (parallel.flat_alias) moves nothing else."""
    k = (P, M, F)
    v = _LAYOUTS.get(k)
    if v is None:
        scratch_f = (L.mgs_backward_scratch_bytes(P, M, F) + 3) // 4
        sizes = [scratch_f, 3 * P, F * P, 3 * P, P, 3 * M * P, 3 * P, 4 * P, 6 * P, 3 * P, 4]
        accum_bytes = ((scratch_f + 3 * P + F * P) * 4 + 15) // 16 * 16  # may reach into the next, fully rewritten, region
        v = _LAYOUTS[k] = (sizes, accum_bytes)
    return v


def _grad_offsets(L, P, M, F):
    """(float offsets of the regions of _grad_layout, total floats)."""
    k = (P, M, F, "offs")
    v = _LAYOUTS.get(k)
    if v is None:
        sizes, _ = _grad_layout(L, P, M, F)
        offs, o = [], 0
        for n in sizes:
            offs.append(o)
            o += n
        v = _LAYOUTS[k] = (tuple(offs), o)
    return v


def compiled():
    """The compiled autograd binding (csrc/mgs_torch.cpp) or None: manigaussian_amd.rasterizer asks it first."""
    return _state.ext()


class use_compiled:
    """`with use_compiled(False):` -- route autograd calls through this ctypes shim (diagnostics that need the Python-side
    ForwardHandle, A/B timing).  No-op when the binding is not built."""

    def __init__(self, on: bool):
        self.on, self.old = bool(on), None

    def __enter__(self):
        e = _state.ext()
        self.old = e.set_enabled(self.on) if e else None
        return self

    def __exit__(self, *exc):
        e = _state.ext()
        if e and self.old is not None:
            e.set_enabled(self.old)


def rasterize_gaussians(background, means3D, colors, language_feature, opacity, scales, rotations, scale_modifier,
                        cov3D_precomp, viewmatrix, projmatrix, tan_fovx, tan_fovy, image_height, image_width, sh,
                        degree, campos, prefiltered, debug, include_feature):
    """RasterizeGaussiansCUDA (RAST/rasterize_points.cu:35-128).
    Returns (num_rendered, out_color [3,H,W], out_language_feature [F,H,W] or [1], radii [P] int32,
             geomBuffer, binningBuffer, imgBuffer).  Like the reference this entry point returns the count as an int, i.e.
    it waits for the preprocess (the autograd path, rasterizer.py, does not)."""
    out = _forward(background, means3D, colors, language_feature, opacity, scales, rotations, scale_modifier,
                   cov3D_precomp, viewmatrix, projmatrix, tan_fovx, tan_fovy, image_height, image_width, sh, degree,
                   campos, prefiltered, debug, include_feature, False, blocking=True)
    return (int(out[0]),) + out[1:7]


def _forward(background, means3D, colors, language_feature, opacity, scales, rotations, scale_modifier, cov3D_precomp,
             viewmatrix, projmatrix, tan_fovx, tan_fovy, image_height, image_width, sh, degree, campos, prefiltered,
             debug, include_feature, want_grad_buffer, blocking=False):
    """rasterize_gaussians + (want_grad_buffer) the backward's allocation, whose accumulator block the forward's
    preprocess kernel zeroes on the side: returns (ForwardHandle or int, color, feature, radii, geom, binning, img,
    grad_buffer or None)."""
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
        language_feature = _aligned16(language_feature)

    with _on_device(dev):
        u8 = dict(dtype=torch.uint8, device=dev)
        if P == 0:  # rasterize_points.cu:92: empty workspaces, zero images
            out_color = torch.zeros((3, H, W), dtype=_F32, device=dev)
            out_feat = torch.zeros((F_user, H, W) if include_feature else (1,), dtype=_F32, device=dev)
            e = torch.empty((0,), **u8)
            return (0, out_color, out_feat, torch.zeros((0,), dtype=torch.int32, device=dev), e, e.clone(), e.clone(),
                    None)
        st = _state.device_state(dev)
        capturing = _capturing()
        if not capturing:
            st.drain()  # reports of earlier forwards that have arrived: learn their counts, raise if one overflowed
        # everything that is a function of the shape alone comes from caches (sizes, the pre-filled argument struct)
        tmpl, opts = _template(P, int(degree), M, F, W, H, float(tan_fovx), float(tan_fovy), float(scale_modifier),
                               bool(prefiltered), bool(debug), include_feature)
        key = (P, W, H, F, opts["tight_bins"])
        T = ((W + 15) // 16) * ((H + 15) // 16)
        cannot_overflow, cap_worst, pool_worst = _worst_case(L, P, W, H, F, T, dev)
        # ... charged against what live forwards of this device already hold (a node keeps its workspace until its backward):
        # V forwards before the first backward take the worst case only while the SUM fits, the rest go by their marks
        worst_bytes = _bin_bytes(L, cap_worst, pool_worst, W, H, F) if cannot_overflow else 0
        if cannot_overflow and not blocking and _state.forward_mode() != "async" and \
                not _state.worst_case_fits(st.index, worst_bytes, dev):
            cannot_overflow = False
        guess = (cap_worst, pool_worst) if cannot_overflow else st.guess(key)  # worst case: no marks, no warm-up call needed
        # prefiltered=True is a checked promise (the reference traps the device): its violation must surface in this call
        lazy = (guess is not None and not blocking and not debug and not prefiltered and
                _state.lazy_allowed(cannot_overflow))
        if capturing and not lazy:
            raise RuntimeError("capturing a rasterizer forward into a HIP graph needs the asynchronous path: "
                               "manigaussian_amd.set_forward_mode('async'), then run this shape eagerly (twice) first so that "
                               "its workspace sizes are known, with debug=False")
        if lazy:
            cap, pool = guess
        else:  # blocking path: the chunk pool is the worst case for the capacity (cannot overflow)
            m = st.marks.get(key)
            cap, pool = (m[0] + m[0] // 4 + 4096 if m else 4 * P + 4096), 0
        # ONE allocation for the three opaque workspaces [geom | img | binning] (each a multiple of 256 bytes), one for the
        # two images; radii and the backward's gradient buffer have lifetimes of their own
        gb, ib = _shape_sizes(L, P, M, W, H)
        # (room for the keys of the direct binning: a worst-case capacity covers them by itself, and the blocking path's carving
        #  is derived from the byte count -- extra bytes would only raise its capacity)
        bb = _bin_bytes(L, cap, pool, W, H, F, P if lazy and cap != cap_worst else 0)
        if _SPLIT_WORKSPACES:
            ws3 = (torch.empty((gb,), **u8), torch.empty((ib,), **u8), torch.empty((bb,), **u8))
            p_geom, p_img, p_bin = ws3[0].data_ptr(), ws3[1].data_ptr(), ws3[2].data_ptr()
            ws = None
        else:
            try:
                ws = torch.empty((gb + ib + bb,), **u8)
            except torch.cuda.OutOfMemoryError:
                if not (lazy and cannot_overflow) or capturing:
                    raise
                # the allocator cannot give the worst case: this forward goes by its marks and waits for the preprocess
                lazy, cannot_overflow = False, False
                m = st.marks.get(key)
                cap, pool = (m[0] + m[0] // 4 + 4096 if m else 4 * P + 4096), 0
                bb = _bin_bytes(L, cap, pool, W, H, F)
                ws = torch.empty((gb + ib + bb,), **u8)
            if lazy and cannot_overflow and want_grad_buffer:
                _state.hold(st.index, worst_bytes, ws)
            p_geom = ws.data_ptr()
            p_img, p_bin = p_geom + gb, p_geom + gb + ib
        if include_feature:
            out = torch.empty((3 + F, H, W), dtype=_F32, device=dev)
            out_color, out_feat = out[:3], out[3:]
        else:
            out_color = torch.empty((3, H, W), dtype=_F32, device=dev)
            out_feat = torch.zeros((1,), dtype=_F32, device=dev)
        radii = torch.empty((P,), dtype=torch.int32, device=dev)  # written for every Gaussian by the preprocess
        a = _lib.MgsRasterArgs.from_buffer_copy(tmpl)
        a.background, a.means3D, a.shs, a.colors_precomp = _ptr(background), means3D.data_ptr(), _ptr(sh), _ptr(colors)
        a.language_feature = language_feature.data_ptr() if include_feature else None
        a.opacities, a.scales, a.rotations, a.cov3D_precomp = _ptr(opacity), _ptr(scales), _ptr(rotations), _ptr(cov3D_precomp)
        a.viewmatrix, a.projmatrix, a.campos = _ptr(viewmatrix), _ptr(projmatrix), _ptr(campos)
        a.geom, a.geom_bytes, a.img, a.img_bytes = p_geom, gb, p_img, ib
        a.binning, a.binning_bytes = p_bin, bb
        m_ = st.marks.get(key)
        if m_ is not None:
            _lib.auto_seg(a, opts, m_[0], T)
        slot_ptr, tag = st.take_slot()
        # The blocking path leaves binning_capacity at 0: the library then derives the carving from the buffer's BYTE COUNT,
        # which is all a caller of the reference-shaped pair rasterize_gaussians / rasterize_gaussians_backward(R: int,
        # binningBuffer) hands back -- forward and backward agree by construction.  The asynchronous path names (capacity,
        # pool) explicitly and its backward reuses this very struct (ForwardHandle.a).
        if lazy:
            a.binning_capacity, a.chunk_pool, a.status_tag, a.async_forward = cap, pool, tag, 1
        else:
            a.status_tag = tag
        grad_buffer = None
        if want_grad_buffer:
            sizes, accum_bytes = _grad_layout(L, P, M, F)
            grad_buffer = torch.empty((sum(sizes),), dtype=_F32, device=dev)
            a.bwd_accum, a.bwd_accum_bytes = grad_buffer.data_ptr(), accum_bytes
        stream = _stream(dev)
        feat_ptr = out_feat.data_ptr() if include_feature else None
        rc, R = _launch_forward(L, a, None, radii, out_color, out_feat, slot_ptr, stream)
        pending = None
        binning2 = None
        if rc == _lib.MGS_NEED_CAPACITY:  # (waiting path) first call for this shape, or the scene grew: bin + render again
            cap = R + R // 4 + 4096  # (R: the reference's 3-sigma-rect count, at least the instances binned)
            binning2 = torch.empty((L.mgs_binning_bytes2(cap, 0, W, H, F),), **u8)
            a.binning, a.binning_bytes, a.binning_capacity, a.chunk_pool = binning2.data_ptr(), binning2.numel(), 0, 0
            rc = L.mgs_rasterize_forward_render(ctypes.byref(a), R, radii.data_ptr(), out_color.data_ptr(), feat_ptr,
                                                stream)
            _lib.check(rc, "rasterize_gaussians")
            st.learn(key, _binned_now(slot_ptr))  # the marks hold BINNED counts (advisor r4: not the 3-sigma-rect count)
        else:
            _lib.check(rc, "rasterize_gaussians")
            # a forward that a backward will follow can be repaired there if it overflowed (recover_forward)
            # (the marks are learned from the device's report -- the instances actually binned --, not from the blocking
            #  call's return value, which is the reference's 3-sigma-rect count)
            pending = _state.Pending(a, 0, slot_ptr, key, captured=capturing, recoverable=want_grad_buffer and not capturing)
            st.add(pending)  # captured forwards report at every replay: _state.check_status()
        handle = ForwardHandle(a, opts, pending, R, (background, means3D, sh, colors, language_feature, opacity, scales,
                                                     rotations, cov3D_precomp, viewmatrix, projmatrix, campos),
                               outs=(_weak(out_color), _weak(out_feat) if include_feature and F == F_user else None))
        if ws is None:
            geom, img, binning = ws3[0], ws3[1], (binning2 if binning2 is not None else ws3[2])
        elif blocking:  # the reference-shaped triple: three tensors whose byte counts describe their carving
            geom, img = ws[:gb], ws[gb:gb + ib]
            binning = binning2 if binning2 is not None else ws[gb + ib:]
        else:         # the autograd path keeps ONE tensor alive (the backward reads the addresses from the handle)
            e = _EMPTY_U8.get(dev)
            if e is None:
                e = _EMPTY_U8[dev] = torch.empty((0,), **u8)
            geom, img, binning = ws, e, (binning2 if binning2 is not None else e)
    if include_feature and F != F_user:
        out_feat = out_feat[:F_user].contiguous()
    return handle, out_color, out_feat, radii, geom, binning, img, grad_buffer


def _settle(handle, radii, dev, count):
    """Backward entry: what is known about this backward's forward?  OK or still running: go on (a report that arrives later
    and says "overflow" raises at the next call into the library: loud, late).  Overflow already known: re-render on the
    blocking path and go on with a complete state (recover_forward).  Anything else (prefiltered violation): raise."""
    p = handle.pending
    if p is None:
        return count
    if p.rc in (_lib.MGS_NEED_CAPACITY, _lib.MGS_RETRY_TABLE_INIT) and not p.captured:
        if _state.overflow_policy() == "raise":
            _state.device_state(dev).drain()  # folds the report into the marks and raises ...
            # ... unless an earlier drain already did:
            if p.rc == _lib.MGS_RETRY_TABLE_INIT:
                raise RuntimeError("the preprocess of this backward's asynchronous forward gave up waiting for its zeroed tile "
                                   "tables and binned nothing: its images are incomplete and the step is lost (overflow policy "
                                   "'raise')")
            raise RuntimeError("the asynchronous rasterizer forward of this backward outgrew its workspace (the scene grew past "
                               "the head-room over earlier calls of its shape): its images are incomplete and the step is "
                               "lost; the next call of the shape gets a larger workspace (overflow policy 'raise')")
        return recover_forward(handle, radii, dev)
    if p.rc not in (_lib.MGS_OK, _lib.MGS_PENDING):
        _state.device_state(dev).drain()
    if p.rc == _lib.MGS_PENDING:
        p.backward_enqueued = True  # too late to repair: if this forward overflowed, the next drain raises
    return count


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
    zeroed by the forward's preprocess kernel) or None.  R: the forward's ForwardHandle (its MgsRasterArgs is reused;
    the count may still be unknown) or the count as an int (buffers sized by mgs_binning_bytes)."""
    L = _lib.lib()
    dev = means3D.device
    P = int(means3D.size(0))
    H, W = int(dL_dout_color.size(1)), int(dL_dout_color.size(2))
    M = int(sh.size(1)) if sh.numel() != 0 else 0
    include_feature = bool(include_feature)
    dL_dout_color = _f32c(dL_dout_color, "dL_dout_color", dev)
    F = F_user = 0
    if include_feature:
        F_user = int(language_feature.size(1))
        F = _padded_F(F_user)
        dL_dout_language_feature = _f32c(dL_dout_language_feature, "dL_dout_language_feature", dev)
        if F != F_user:
            dL_dout_language_feature = torch.cat(
                [dL_dout_language_feature, dL_dout_language_feature.new_zeros((F - F_user, H, W))], 0)
    handle = R if isinstance(R, ForwardHandle) else None
    keep = None
    with _on_device(dev):
        if P == 0:
            z = lambda *s_: torch.zeros(s_, dtype=_F32, device=dev)  # noqa: E731
            return (z(0, 3), z(0, 3), z(0, F_user) if include_feature else z(1), z(0, 1), z(0, 3), z(0, 6),
                    z(0, M, 3), z(0, 3), z(0, 4))
        # ONE allocation for the scratch accumulators and every gradient: the three regions the render backward
        # accumulates into (acc8 | dL_dcolors | dL_dfeature) come first and are contiguous, so the library zeroes
        # them with a single fill; everything else is fully written by the kernels.  Regions are addressed by offset
        # (no split / view tensors on the way in; one as_strided per gradient on the way out).
        (o_scr, o_col, o_feat, o_m3, o_op, o_sh, o_sc, o_rot, o_cov, o_m2, _o_pad), total = _grad_offsets(L, P, M, F)
        prezeroed = grad_buffer is not None and grad_buffer.numel() == total
        flat = grad_buffer if prezeroed else torch.empty((total,), dtype=_F32, device=dev)
        base = flat.data_ptr()
        if handle is not None:  # the forward's arguments, as they were (same tensors: they are saved in the autograd ctx)
            a = handle.a
            count = handle.num_rendered_nowait()
            count = _settle(handle, radii, dev, count)
        else:
            means3D = _f32c(means3D, "means3D", dev)
            colors = _f32c(colors, "colors_precomp", dev)
            scales = _f32c(scales, "scales", dev)
            rotations = _f32c(rotations, "rotations", dev)
            cov3D_precomp = _f32c(cov3D_precomp, "cov3D_precomp", dev)
            sh = _f32c(sh, "sh", dev)
            if include_feature:
                language_feature = _f32c(language_feature, "language_feature_precomp", dev)
                if F != F_user:
                    language_feature = torch.nn.functional.pad(language_feature, (0, F - F_user))
                language_feature = _aligned16(language_feature)
            keep = (means3D, colors, scales, rotations, cov3D_precomp, sh, language_feature)
            a = _lib.MgsRasterArgs()
            _fill_args(a, P=P, D=int(degree), M=M, F=F, W=W, H=H, tanfovx=float(tan_fovx), tanfovy=float(tan_fovy),
                       scale_modifier=float(scale_modifier), prefiltered=False, debug=debug,
                       include_feature=include_feature, background=_f32c(background, "background", dev),
                       means3D=means3D, sh=sh, colors=colors, language_feature=language_feature, opacity=None,
                       scales=scales, rotations=rotations, cov3D_precomp=cov3D_precomp,
                       viewmatrix=_f32c(viewmatrix, "viewmatrix", dev), projmatrix=_f32c(projmatrix, "projmatrix", dev),
                       campos=_f32c(campos, "campos", dev), geom=geomBuffer, binning=binningBuffer, img=imageBuffer)
            _lib.fill_options(a)
            count = int(R)
        a.accum_prezeroed = 1 if prezeroed else 0
        _lib.check(L.mgs_rasterize_backward(
            ctypes.byref(a), count, radii.data_ptr(), dL_dout_color.data_ptr(),
            _ptr(dL_dout_language_feature) if include_feature else None, base + 4 * o_m2, None,
            base + 4 * o_op, base + 4 * o_col, (base + 4 * o_feat) if include_feature else None,
            base + 4 * o_m3, base + 4 * o_cov, (base + 4 * o_sh) if M else None, base + 4 * o_sc, base + 4 * o_rot,
            base + 4 * o_scr, (o_col - o_scr) * 4, _stream(dev)), "rasterize_gaussians_backward")
        view = flat.as_strided
        g_feat = view((P, F), (F, 1), o_feat) if include_feature else torch.zeros((1,), dtype=_F32, device=dev)
        g_sh = view((P, M, 3), (3 * M, 3, 1), o_sh) if M else flat.new_empty((P, 0, 3))
        out = (view((P, 3), (3, 1), o_m2), view((P, 3), (3, 1), o_col), g_feat, view((P, 1), (1, 1), o_op),
               view((P, 3), (3, 1), o_m3), view((P, 6), (6, 1), o_cov), g_sh, view((P, 3), (3, 1), o_sc),
               view((P, 4), (4, 1), o_rot))
    del keep
    if include_feature and F != F_user:
        out = out[:2] + (out[2][:, :F_user].contiguous(),) + out[3:]
    return out


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
