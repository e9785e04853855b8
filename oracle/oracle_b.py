"""ctypes front-end for oracle/mgs_oracle.c (Oracle B, the CPU restatement of the reference).

TEST INFRASTRUCTURE ONLY -- imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg, never by the product path.  Pinned against outputs of the reference's own
kernels (tests/golden/ref, made by oracle/_ref = the reference sources built with hipcc; see
mgs_oracle.c's header), and by oracle_a.py + closed forms.

Mirrors the call shape of RAST/diff_gaussian_rasterization/__init__.py:_RasterizeGaussians:
forward(...) -> (color, feature, radii, state); backward(state, dL_dcolor, dL_dfeat) -> grads
in the order the reference returns them (__init__.py:151-162).
"""
import ctypes
import os
import subprocess

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libmgs_oracle.so")
_lib = None

c_fp = ctypes.POINTER(ctypes.c_float)
c_ip = ctypes.POINTER(ctypes.c_int)
c_u8p = ctypes.POINTER(ctypes.c_uint8)
c_u32p = ctypes.POINTER(ctypes.c_uint32)
c_u64p = ctypes.POINTER(ctypes.c_uint64)


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "mgs_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libmgs_oracle.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = ctypes.CDLL(_LIB_PATH)
        L.orc_forward.restype = ctypes.c_void_p
        L.orc_forward.argtypes = [ctypes.c_int] * 4 + [c_fp, ctypes.c_int, ctypes.c_int] + [c_fp] * 6 + [
            ctypes.c_float, c_fp, c_fp, c_fp, c_fp, c_fp, ctypes.c_float, ctypes.c_float, ctypes.c_int,
            ctypes.c_int, c_fp, c_fp, c_ip, c_ip]
        L.orc_backward.restype = None
        L.orc_backward.argtypes = [ctypes.c_void_p] + [c_fp] * 12
        L.orc_free.restype = None
        L.orc_free.argtypes = [ctypes.c_void_p]
        L.orc_set_threads.argtypes = [ctypes.c_int]
        L.orc_max_threads.restype = ctypes.c_int
        L.orc_mark_visible.argtypes = [ctypes.c_int, c_fp, c_fp, c_fp, c_u8p]
        L.orc_get_higher_msb.restype = ctypes.c_uint32
        L.orc_get_higher_msb.argtypes = [ctypes.c_uint32]
        L.orc_num_rendered.restype = ctypes.c_int
        L.orc_num_rendered.argtypes = [ctypes.c_void_p]
        for name, rt in [("depths", c_fp), ("means2D", c_fp), ("conic_opacity", c_fp), ("rgb", c_fp),
                         ("cov3D", c_fp), ("clamped", c_u8p), ("tiles_touched", c_u32p),
                         ("point_list", c_u32p), ("keys", c_u64p), ("ranges", c_u32p),
                         ("final_T", c_fp), ("n_contrib", c_u32p)]:
            fn = getattr(L, "orc_" + name)
            fn.restype = rt
            fn.argtypes = [ctypes.c_void_p]
        L.orc_fragile_mask.restype = None
        L.orc_fragile_mask.argtypes = [ctypes.c_void_p, ctypes.c_float, c_u8p, c_u8p]
        L.orc_deform_apply_fwd.argtypes = [ctypes.c_int] + [c_fp] * 5
        L.orc_deform_apply_bwd.argtypes = [ctypes.c_int] + [c_fp] * 5
        _lib = L
    return _lib


def _f32(t):
    """-> (contiguous float32 numpy array or None, ctypes pointer or NULL)."""
    if t is None:
        return None, None
    if isinstance(t, torch.Tensor):
        if t.numel() == 0:
            return None, None
        a = t.detach().cpu().contiguous().float().numpy()
    else:
        a = np.ascontiguousarray(t, dtype=np.float32)
        if a.size == 0:
            return None, None
    return a, a.ctypes.data_as(c_fp)


class State:
    """Owns the OrcState handle and keeps the borrowed input arrays alive."""

    def __init__(self, handle, keep, P, M, F, W, H, include_feature):
        self.handle, self.keep = handle, keep
        self.P, self.M, self.F, self.W, self.H, self.include_feature = P, M, F, W, H, include_feature

    def __del__(self):
        h, self.handle = getattr(self, "handle", None), None
        if h and _lib is not None:
            try:
                _lib.orc_free(h)
            except Exception:  # interpreter shutdown
                pass

    @property
    def num_rendered(self):
        return lib().orc_num_rendered(self.handle)

    def array(self, name):
        L = lib()
        P, T = self.P, ((self.W + 15) // 16) * ((self.H + 15) // 16)
        R, N = self.num_rendered, self.W * self.H
        shape = {"depths": (P,), "means2D": (P, 2), "conic_opacity": (P, 4), "rgb": (P, 3), "cov3D": (P, 6),
                 "clamped": (P, 3), "tiles_touched": (P,), "point_list": (R,), "keys": (R,),
                 "ranges": (T, 2), "final_T": (N,), "n_contrib": (N,)}[name]
        n = int(np.prod(shape))
        if n == 0:
            return np.zeros(shape)
        ptr = getattr(L, "orc_" + name)(self.handle)
        return np.ctypeslib.as_array(ptr, shape=(n,)).reshape(shape).copy()


def fragile_mask(st: State, rel_eps: float = 2e-5):
    """bool [H,W]: pixels whose walk passes within rel_eps of a hard threshold (see orc_fragile_mask)."""
    m = np.zeros((st.H * st.W,), np.uint8)
    lib().orc_fragile_mask(st.handle, float(rel_eps), m.ctypes.data_as(c_u8p), None)
    return torch.from_numpy(m.reshape(st.H, st.W).astype(bool))


def fragile_gaussians(st: State, rel_eps: float = 2e-5):
    """bool [P]: Gaussians with at least one (pixel, Gaussian) pair within rel_eps of a hard threshold."""
    m = np.zeros((st.H * st.W,), np.uint8)
    g = np.zeros((max(st.P, 1),), np.uint8)
    lib().orc_fragile_mask(st.handle, float(rel_eps), m.ctypes.data_as(c_u8p), g.ctypes.data_as(c_u8p))
    return torch.from_numpy(g[:st.P].astype(bool))


def set_threads(n: int):
    lib().orc_set_threads(int(n))


def max_threads() -> int:
    return lib().orc_max_threads()


def forward(means3D, opacities, settings, shs=None, colors_precomp=None, language_feature=None,
            scales=None, rotations=None, cov3D_precomp=None):
    """settings: any object with the 13 GaussianRasterizationSettings fields."""
    L = lib()
    P = int(means3D.shape[0])
    keep = []
    def p(t):
        a, ptr = _f32(t)
        keep.append(a)
        return ptr
    M = int(shs.shape[1]) if (shs is not None and shs.numel() != 0) else 0
    inc = bool(settings.include_feature)
    F = int(language_feature.shape[1]) if (language_feature is not None and language_feature.numel() != 0) else 0
    H, W = int(settings.image_height), int(settings.image_width)
    color = np.zeros((3, H, W), np.float32)
    feat = np.zeros((F, H, W), np.float32) if inc else np.zeros((1,), np.float32)
    radii = np.zeros((max(P, 1),), np.int32)
    nr = ctypes.c_int(0)
    h = L.orc_forward(P, int(settings.sh_degree), M, F, p(settings.bg), W, H, p(means3D), p(shs),
                      p(colors_precomp), p(language_feature), p(opacities), p(scales),
                      float(settings.scale_modifier), p(rotations), p(cov3D_precomp), p(settings.viewmatrix),
                      p(settings.projmatrix), p(settings.campos), float(settings.tanfovx),
                      float(settings.tanfovy), int(bool(settings.prefiltered)), int(inc),
                      color.ctypes.data_as(c_fp), feat.ctypes.data_as(c_fp), radii.ctypes.data_as(c_ip),
                      ctypes.byref(nr))
    if not h:
        raise RuntimeError("oracle forward failed (prefiltered trap or unsupported F)")
    st = State(h, keep, P, M, F, W, H, inc)
    return torch.from_numpy(color), torch.from_numpy(feat), torch.from_numpy(radii[:P]), st


def backward(st: State, dL_dcolor, dL_dfeat=None):
    L = lib()
    P, M, F = st.P, st.M, st.F
    a_c, p_c = _f32(dL_dcolor)
    a_f, p_f = _f32(dL_dfeat if st.include_feature else None)
    z = lambda *s: np.zeros(s if int(np.prod(s)) > 0 else (1,), np.float32)
    g = dict(means2D=z(P, 3), conic=z(P, 2, 2), opacities=z(P, 1), colors_precomp=z(P, 3),
             language_feature=z(P, max(F, 1)), means3D=z(P, 3), cov3D=z(P, 6), sh=z(P, max(M, 1), 3),
             scales=z(P, 3), rotations=z(P, 4))
    ptr = lambda k: g[k].ctypes.data_as(c_fp)
    L.orc_backward(st.handle, p_c, p_f, ptr("means2D"), ptr("conic"), ptr("opacities"), ptr("colors_precomp"),
                   ptr("language_feature"), ptr("means3D"), ptr("cov3D"), ptr("sh"), ptr("scales"),
                   ptr("rotations"))
    if M == 0:
        g["sh"] = np.zeros((P, 0, 3), np.float32)
    if not st.include_feature or F == 0:
        g["language_feature"] = np.zeros((1,), np.float32)
    return {k: torch.from_numpy(v) for k, v in g.items()}


def mark_visible(positions, viewmatrix, projmatrix):
    L = lib()
    P = int(positions.shape[0])
    a, pa = _f32(positions)
    v, pv = _f32(viewmatrix)
    pr, ppr = _f32(projmatrix)
    out = np.zeros((max(P, 1),), np.uint8)
    if P:
        L.orc_mark_visible(P, pa, pv, ppr, out.ctypes.data_as(c_u8p))
    return torch.from_numpy(out[:P].astype(bool))


def get_higher_msb(n: int) -> int:
    return int(lib().orc_get_higher_msb(int(n)))


def deform_apply_fwd(xyz, rot, delta):
    L = lib()
    N = int(xyz.shape[0])
    a, pa = _f32(xyz); b, pb = _f32(rot); d, pd = _f32(delta)
    xo = np.zeros((N, 3), np.float32); ro = np.zeros((N, 4), np.float32)
    L.orc_deform_apply_fwd(N, pa, pb, pd, xo.ctypes.data_as(c_fp), ro.ctypes.data_as(c_fp))
    return torch.from_numpy(xo), torch.from_numpy(ro)


def deform_apply_bwd(rot, delta, g_xyz, g_rot):
    L = lib()
    N = int(rot.shape[0])
    b, pb = _f32(rot); d, pd = _f32(delta); gx, pgx = _f32(g_xyz); gr, pgr = _f32(g_rot)
    gd = np.zeros((N, 7), np.float32)
    L.orc_deform_apply_bwd(N, pb, pd, pgx, pgr, gd.ctypes.data_as(c_fp))
    return torch.from_numpy(gd)
