"""oracle/ref_cuda.py -- TEST INFRASTRUCTURE ONLY (never imported by manigaussian_amd/).

ctypes front-end of oracle/_ref/libmgs_ref.so: the REFERENCE's own rasterizer
(RAST/cuda_rasterizer/{forward,backward,rasterizer_impl}.cu, RAST = third_party/gaussian-splatting/submodules/
diff-gaussian-rasterization) compiled unmodified with hipcc through oracle/refshim (oracle/Makefile, target _ref).  It
exists to PIN the other two oracles and the HIP path against what the reference itself computes:

  * tests/golden/make_golden_ref.py runs it on the GPU box and writes tests/golden/ref/*.npz (committed);
  * tests/test_gpu_parity.py::test_live_reference compares the HIP path with it directly when the .so is present.

What is and is not "the reference" here: every kernel and the host orchestration (CudaRasterizer::Rasterizer::forward /
backward, rasterizer_impl.cu:198-463) are the reference's sources, byte for byte, built for gfx950.  Three things are
substitutes: glm (an un-vendored submodule, restated in refshim/glm/glm.hpp: vec3/vec4/mat3, column-major), CUB ->
hipCUB (same radix sort / scan contracts) and the CUDA runtime names -> HIP.  The feature width is the reference build's
NUM_CHANNELS_language_feature = 3 (RAST/cuda_rasterizer/config.h:16).
"""
import ctypes
import os

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# stock build (3 feature channels, config.h:16) and the same sources rebuilt at 32 channels (oracle/Makefile)
LIB_PATHS = {3: os.path.join(_HERE, "_ref", "libmgs_ref.so"), 32: os.path.join(_HERE, "_ref", "libmgs_ref_f32.so"),
             8: os.path.join(_HERE, "_ref", "libmgs_ref_f8.so")}
LIB_PATH = LIB_PATHS[3]
_libs = {}
c_fp = ctypes.POINTER(ctypes.c_float)
c_ip = ctypes.POINTER(ctypes.c_int)


def available(F: int = 3) -> bool:
    return F in LIB_PATHS and os.path.exists(LIB_PATHS[F]) and torch.cuda.is_available()


_INPUTS = ([ctypes.c_int] * 5 + [c_fp] * 7 + [ctypes.c_float] + [c_fp] * 5 + [ctypes.c_float] * 2 + [ctypes.c_int] +
           [c_fp] * 2)


def lib(F: int = 3):
    if F not in _libs:
        L = ctypes.CDLL(LIB_PATHS[F])
        L.ref_num_feature_channels.restype = ctypes.c_int
        L.ref_forward_backward.restype = ctypes.c_int
        L.ref_forward_backward.argtypes = _INPUTS + [c_fp, c_fp, c_ip] + [c_fp] * 9
        L.ref_forward_geometry.restype = ctypes.c_int
        L.ref_forward_geometry.argtypes = _INPUTS + [c_ip] + [c_fp] * 5
        L.ref_bench.restype = ctypes.c_int
        L.ref_bench.argtypes = _INPUTS + [ctypes.c_int, ctypes.c_int] + [c_fp] * 3
        assert int(L.ref_num_feature_channels()) == F
        _libs[F] = L
    return _libs[F]


def _f32(t):
    if t is None:
        return None, None
    a = np.ascontiguousarray(t.detach().cpu().numpy() if isinstance(t, torch.Tensor) else np.asarray(t), np.float32)
    if a.size == 0:
        return None, None
    return a, a.ctypes.data_as(c_fp)


def _width(language_feature) -> int:
    """Which build serves this call: the feature width is a compile-time constant of the reference."""
    F = int(language_feature.shape[1]) if (language_feature is not None and language_feature.numel() != 0) else 3
    if F not in LIB_PATHS:
        raise ValueError(f"no reference build for {F} feature channels (have {sorted(LIB_PATHS)})")
    return F


def forward_backward(means3D, opacities, settings, d_color, d_feat=None, shs=None, colors_precomp=None,
                     language_feature=None, scales=None, rotations=None, cov3D_precomp=None):
    """One forward + backward through the reference kernels.  Returns (color, feat, radii, grads, num_rendered) with
    the gradient names of oracle_b.backward."""
    F = _width(language_feature)
    L = lib(F)
    P = int(means3D.shape[0])
    M = int(shs.shape[1]) if (shs is not None and shs.numel() != 0) else 0
    inc = bool(settings.include_feature)
    H, W = int(settings.image_height), int(settings.image_width)
    if language_feature is None or language_feature.numel() == 0:
        language_feature = torch.zeros(P, F)
    keep = []

    def p(t):
        a, ptr = _f32(t)
        keep.append(a)
        return ptr
    z = lambda *s: np.zeros(s, np.float32)
    color, feat, radii = z(3, H, W), z(F, H, W), np.zeros((max(P, 1),), np.int32)
    g = dict(means2D=z(P, 3), opacities=z(P, 1), colors_precomp=z(P, 3), language_feature=z(P, F), means3D=z(P, 3),
             cov3D=z(P, 6), sh=z(P, max(M, 1), 3), scales=z(P, 3), rotations=z(P, 4))
    o = lambda a: a.ctypes.data_as(c_fp)
    R = L.ref_forward_backward(
        P, int(settings.sh_degree), M, W, H, p(settings.bg), p(means3D), p(shs), p(colors_precomp), p(language_feature),
        p(opacities), p(scales), float(settings.scale_modifier), p(rotations), p(cov3D_precomp), p(settings.viewmatrix),
        p(settings.projmatrix), p(settings.campos), float(settings.tanfovx), float(settings.tanfovy), int(inc),
        p(d_color), p(d_feat if inc else None), o(color), o(feat), radii.ctypes.data_as(c_ip), o(g["means2D"]),
        o(g["opacities"]), o(g["colors_precomp"]), o(g["language_feature"]), o(g["means3D"]), o(g["cov3D"]), o(g["sh"]),
        o(g["scales"]), o(g["rotations"]))
    if R < 0:
        raise RuntimeError(f"reference rasterizer failed ({R})")
    if M == 0:
        g["sh"] = np.zeros((P, 0, 3), np.float32)
    t = torch.from_numpy
    return t(color), t(feat), t(radii[:P]), {k: t(v) for k, v in g.items()}, int(R)


def forward_geometry(means3D, opacities, settings, shs=None, colors_precomp=None, language_feature=None, scales=None,
                     rotations=None, cov3D_precomp=None):
    """What the reference's preprocess (forward.cu:156-257) left in its GeometryState after one forward: dict(radii,
    means2D [P,2], conic_opacity [P,4], depths [P], rgb [P,3], cov3D [P,6]) as numpy arrays; rows with radii == 0 are
    unspecified."""
    F = _width(language_feature)
    L = lib(F)
    P = int(means3D.shape[0])
    M = int(shs.shape[1]) if (shs is not None and shs.numel() != 0) else 0
    if language_feature is None or language_feature.numel() == 0:
        language_feature = torch.zeros(P, F)
    keep = []

    def p(t):
        a, ptr = _f32(t)
        keep.append(a)
        return ptr
    z = lambda *s: np.zeros(s, np.float32)
    out = dict(radii=np.zeros((max(P, 1),), np.int32), means2D=z(P, 2), conic_opacity=z(P, 4), depths=z(P), rgb=z(P, 3),
               cov3D=z(P, 6))
    o = lambda a: a.ctypes.data_as(c_fp)
    R = L.ref_forward_geometry(
        P, int(settings.sh_degree), M, int(settings.image_width), int(settings.image_height), p(settings.bg), p(means3D),
        p(shs), p(colors_precomp), p(language_feature), p(opacities), p(scales), float(settings.scale_modifier),
        p(rotations), p(cov3D_precomp), p(settings.viewmatrix), p(settings.projmatrix), p(settings.campos),
        float(settings.tanfovx), float(settings.tanfovy), int(bool(settings.include_feature)), None, None,
        out["radii"].ctypes.data_as(c_ip), o(out["means2D"]), o(out["conic_opacity"]), o(out["depths"]), o(out["rgb"]),
        o(out["cov3D"]))
    if R < 0:
        raise RuntimeError(f"reference rasterizer failed ({R})")
    out["radii"] = out["radii"][:P]
    out["num_rendered"] = int(R)
    return out


def bench(means3D, opacities, settings, d_color, d_feat, warmup=10, iters=50, shs=None, colors_precomp=None,
          language_feature=None, scales=None, rotations=None, cov3D_precomp=None):
    """Times the reference kernels on this GPU, inputs resident (ref_wrapper.cu:ref_bench).  Returns
    dict(ms_step, ms_fwd, ms_bwd, num_rendered): per forward+backward pass."""
    F = _width(language_feature)
    L = lib(F)
    P = int(means3D.shape[0])
    M = int(shs.shape[1]) if (shs is not None and shs.numel() != 0) else 0
    keep = []

    def p(t):
        a, ptr = _f32(t)
        keep.append(a)
        return ptr
    out = (ctypes.c_float * 3)()
    ptr = lambda i: ctypes.cast(ctypes.byref(out, 4 * i), c_fp)
    R = L.ref_bench(
        P, int(settings.sh_degree), M, int(settings.image_width), int(settings.image_height), p(settings.bg), p(means3D),
        p(shs), p(colors_precomp), p(language_feature), p(opacities), p(scales), float(settings.scale_modifier),
        p(rotations), p(cov3D_precomp), p(settings.viewmatrix), p(settings.projmatrix), p(settings.campos),
        float(settings.tanfovx), float(settings.tanfovy), int(bool(settings.include_feature)), p(d_color), p(d_feat),
        int(warmup), int(iters), ptr(0), ptr(1), ptr(2))
    if R < 0:
        raise RuntimeError(f"reference bench failed ({R})")
    return dict(ms_step=out[0] / iters, ms_fwd=out[1] / iters, ms_bwd=out[2] / iters, num_rendered=int(R))
