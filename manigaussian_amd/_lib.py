"""ctypes binding of libmgsplat.so (C ABI declared in include/mgsplat.h).

There is no CPU fallback and no other backend: if the HIP library is missing or stale this module
raises, loudly, at import of the ops (build it with `python -c "import __graft_entry__ as g; g.build()"`
or `make -C manigaussian_amd/csrc`).
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libmgsplat.so")
ABI_VERSION = 8

c_fp = ctypes.c_void_p  # device pointers travel as integers (tensor.data_ptr())
c_i32 = ctypes.c_int32
c_sz = ctypes.c_size_t

# error codes (include/mgsplat.h)
MGS_OK, MGS_ERR_INVALID_ARG, MGS_ERR_HIP, MGS_ERR_WORKSPACE, MGS_ERR_NON_RGB = 0, -1, -2, -3, -4
MGS_NEED_CAPACITY = 1
MGS_PENDING = 2
MGS_RETRY_TABLE_INIT = 3  # mgs_forward_result: the preprocess's table hand-shake gave up; re-run with table_init = 1

SUPPORTED_F = (3, 4, 8, 16, 32, 64)


class MgsOptions(ctypes.Structure):
    _fields_ = [("set", c_i32), ("tight_bins", c_i32), ("fast_exp", c_i32), ("exact_cull", c_i32), ("bin_mode", c_i32),
                ("seg", c_i32), ("gm_waves", c_i32), ("dbg", c_i32), ("table_init", c_i32)]


class MgsRasterArgs(ctypes.Structure):
    _fields_ = [
        ("P", c_i32), ("D", c_i32), ("M", c_i32), ("F", c_i32), ("W", c_i32), ("H", c_i32),
        ("tanfovx", ctypes.c_float), ("tanfovy", ctypes.c_float), ("scale_modifier", ctypes.c_float),
        ("prefiltered", c_i32), ("debug", c_i32), ("include_feature", c_i32),
        ("background", c_fp), ("means3D", c_fp), ("shs", c_fp), ("colors_precomp", c_fp),
        ("language_feature", c_fp), ("opacities", c_fp), ("scales", c_fp), ("rotations", c_fp),
        ("cov3D_precomp", c_fp), ("viewmatrix", c_fp), ("projmatrix", c_fp), ("campos", c_fp),
        ("geom", c_fp), ("geom_bytes", c_sz), ("binning", c_fp), ("binning_bytes", c_sz),
        ("img", c_fp), ("img_bytes", c_sz),
        ("bwd_accum", c_fp), ("bwd_accum_bytes", c_sz), ("accum_prezeroed", c_i32),
        ("binning_capacity", c_i32), ("chunk_pool", c_i32), ("status_tag", ctypes.c_uint32), ("async_forward", c_i32),
        ("opt", MgsOptions),
    ]


class MgsView(ctypes.Structure):
    _fields_ = [("tanfovx", ctypes.c_float), ("tanfovy", ctypes.c_float), ("viewmatrix", c_fp), ("projmatrix", c_fp),
                ("campos", c_fp)]


MAX_VIEWS = 16

_EXPORTS = {
    # name: (restype, argtypes)
    "mgs_abi_version": (ctypes.c_int, []),
    "mgs_last_error": (ctypes.c_char_p, []),
    "mgs_build_id": (ctypes.c_char_p, []),
    "mgs_options_default": (None, [ctypes.POINTER(MgsOptions)]),
    "mgs_set_option": (ctypes.c_int, [ctypes.c_char_p, ctypes.c_int]),
    "mgs_get_option": (ctypes.c_int, [ctypes.c_char_p]),
    "mgs_geom_bytes": (c_sz, [ctypes.c_int] * 4),
    "mgs_img_bytes": (c_sz, [ctypes.c_int, ctypes.c_int]),
    "mgs_binning_bytes": (c_sz, [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int]),
    "mgs_binning_bytes2": (c_sz, [ctypes.c_int] * 5),
    "mgs_binning_direct_extra": (c_sz, [ctypes.c_int] * 4),
    "mgs_chunk_pool_max": (ctypes.c_int, [ctypes.c_int] * 3),
    "mgs_forward_result": (ctypes.c_int, [ctypes.POINTER(MgsRasterArgs), c_fp, ctypes.POINTER(c_i32),
                                          ctypes.POINTER(c_i32), ctypes.POINTER(c_i32)]),
    "mgs_forward_result_views": (ctypes.c_int, [ctypes.POINTER(MgsRasterArgs), c_i32, c_fp, ctypes.POINTER(c_i32),
                                                ctypes.POINTER(c_i32), ctypes.POINTER(c_i32)]),
    "mgs_views_binning_bytes2": (c_sz, [ctypes.c_int] * 6),
    "mgs_views_chunk_pool_max": (ctypes.c_int, [ctypes.c_int] * 4),
    "mgs_calibration_kernel": (ctypes.c_int, [ctypes.c_int, c_fp, c_fp]),
    "mgs_backward_scratch_bytes": (c_sz, [ctypes.c_int, ctypes.c_int, ctypes.c_int]),
    "mgs_rasterize_forward_preprocess": (ctypes.c_int, [ctypes.POINTER(MgsRasterArgs), c_fp,
                                                        ctypes.POINTER(c_i32), c_fp]),
    "mgs_rasterize_forward_render": (ctypes.c_int, [ctypes.POINTER(MgsRasterArgs), c_i32, c_fp, c_fp, c_fp, c_fp]),
    "mgs_rasterize_forward": (ctypes.c_int, [ctypes.POINTER(MgsRasterArgs), c_fp, c_fp, c_fp, ctypes.POINTER(c_i32),
                                             c_fp, c_fp]),
    "mgs_rasterize_backward": (ctypes.c_int, [ctypes.POINTER(MgsRasterArgs), c_i32, c_fp] + [c_fp] * 12 +
                               [c_fp, c_sz, c_fp]),
    "mgs_views_geom_bytes": (c_sz, [ctypes.c_int] * 5),
    "mgs_views_img_bytes": (c_sz, [ctypes.c_int] * 3),
    "mgs_views_binning_bytes": (c_sz, [ctypes.c_int] * 5),
    "mgs_views_backward_scratch_bytes": (c_sz, [ctypes.c_int] * 4),
    "mgs_rasterize_forward_views": (ctypes.c_int, [ctypes.POINTER(MgsRasterArgs), c_i32, ctypes.POINTER(MgsView), c_fp,
                                                   c_fp, c_fp, ctypes.POINTER(c_i32), c_fp, c_fp]),
    "mgs_rasterize_backward_views": (ctypes.c_int, [ctypes.POINTER(MgsRasterArgs), c_i32, ctypes.POINTER(MgsView), c_i32,
                                                    c_fp] + [c_fp] * 12 + [c_fp, c_sz, c_fp]),
    "mgs_mark_visible": (ctypes.c_int, [ctypes.c_int, c_fp, c_fp, c_fp, c_fp, c_fp]),
    "mgs_deform_assemble_forward": (ctypes.c_int, [ctypes.c_int] * 4 + [c_fp] * 10 + [c_fp]),
    "mgs_deform_assemble_backward": (ctypes.c_int, [ctypes.c_int] * 5 + [c_fp] * 3 + [c_fp]),
    "mgs_deform_apply_forward": (ctypes.c_int, [ctypes.c_int] + [c_fp] * 5 + [c_fp]),
    "mgs_deform_apply_backward": (ctypes.c_int, [ctypes.c_int] + [c_fp] * 5 + [c_fp]),
    "mgs_mlp_relu_bias": (ctypes.c_int, [ctypes.c_int] * 2 + [c_fp] * 4 + [c_fp]),
    "mgs_mlp_relu_backward": (ctypes.c_int, [ctypes.c_int] * 2 + [c_fp] * 5 + [c_fp]),
    "mgs_regress_epilogue_forward": (ctypes.c_int, [ctypes.c_int] + [c_fp] * 9 + [c_fp]),
    "mgs_regress_epilogue_backward": (ctypes.c_int, [ctypes.c_int] + [c_fp] * 9 + [c_fp]),
    "mgs_voxel_sample_pe_forward": (ctypes.c_int, [ctypes.c_int] * 6 + [ctypes.c_float, ctypes.POINTER(ctypes.c_float)] +
                                    [c_fp] * 3 + [c_fp]),
    "mgs_voxel_sample_backward": (ctypes.c_int, [ctypes.c_int] * 5 + [ctypes.POINTER(ctypes.c_float)] + [c_fp] * 2 +
                                  [ctypes.c_int, c_fp, c_fp]),
    "mgs_novel_calib": (ctypes.c_int, [ctypes.c_int, c_fp, c_fp, ctypes.c_int, ctypes.c_int] + [ctypes.c_float] * 6 +
                        [c_fp] * 5 + [c_fp, c_fp]),
    "mgs_novel_calib_host": (ctypes.c_int, [ctypes.c_int, c_fp, c_fp, ctypes.c_int, ctypes.c_int] + [ctypes.c_float] * 6 +
                             [c_fp] * 5),
    "mgs_forward_stats": (ctypes.c_int, [ctypes.POINTER(MgsRasterArgs), c_i32, ctypes.POINTER(ctypes.c_int64),
                                         ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_int64), c_fp]),
    "mgs_debug_geom_layout": (ctypes.c_int, [ctypes.c_int] * 4 + [ctypes.POINTER(c_sz)] * 4),
    "mgs_debug_binning_layout": (ctypes.c_int, [ctypes.POINTER(MgsRasterArgs), c_i32] + [ctypes.POINTER(c_sz)] * 3 +
                                 [ctypes.POINTER(c_i32)]),
    "mgs_debug_direct_keys": (ctypes.c_int, [ctypes.POINTER(MgsRasterArgs), c_i32, ctypes.POINTER(c_sz), ctypes.POINTER(c_i32)]),
    "mgs_debug_read_trace": (ctypes.c_int, [ctypes.c_void_p, c_sz]),
    "mgs_debug_read_trace_bwd": (ctypes.c_int, [ctypes.c_void_p, c_sz]),
    "mgs_debug_read_trace_bin": (ctypes.c_int, [ctypes.c_void_p, c_sz]),
    "mgs_selftest": (ctypes.c_int, [c_fp]),
    "mgs_profile_num_stages": (ctypes.c_int, []),
    "mgs_profile_stage_name": (ctypes.c_char_p, [ctypes.c_int]),
    "mgs_profile_read": (ctypes.c_int, [ctypes.POINTER(ctypes.c_double), ctypes.POINTER(c_i32), ctypes.c_int]),
}

_lib = None


def exported_symbols():
    """Every entry point include/mgsplat.h declares."""
    return sorted(_EXPORTS)


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} not found: the HIP library is not built. There is no fallback path; run "
                "`make -C manigaussian_amd/csrc` (hipcc, gfx950) or __graft_entry__.build().")
        L = ctypes.CDLL(LIB_PATH)
        for name, (rt, at) in _EXPORTS.items():
            fn = getattr(L, name)  # AttributeError if the .so is stale
            fn.restype, fn.argtypes = rt, at
        v = L.mgs_abi_version()
        if v != ABI_VERSION:
            raise ImportError(f"libmgsplat ABI version {v} != expected {ABI_VERSION}; rebuild")
        _lib = L
    return _lib


def build_id() -> str:
    """Hash of the sources libmgsplat.so was compiled from (baked in at build time)."""
    return lib().mgs_build_id().decode()


def last_error() -> str:
    return lib().mgs_last_error().decode("utf-8", "replace")


def check(rc: int, what: str):
    if rc != 0:
        raise RuntimeError(f"{what}: {last_error()} (code {rc})")


# Per-call tuning switches (MgsOptions): the C ABI has no process-wide option state; this dict is merely the DEFAULT the
# Python shim copies into every call's MgsRasterArgs.opt (a forward's values travel to its backward in the autograd ctx).
DEFAULT_OPTIONS = dict(tight_bins=1, fast_exp=0, exact_cull=1, bin_mode=2, seg=2048, gm_waves=12, dbg=0, table_init=0)
OPTIONS_VERSION = [0]  # bumped by set_option: callers that cache a filled MgsOptions key it on this


def set_option(key: str, value: int):
    """"profile": the library's one process-wide DIAGNOSTIC switch (stage timers).  Any other key: the default of the
    per-call option of that name for calls made from this process afterwards."""
    if key == "profile":
        check(lib().mgs_set_option(key.encode(), int(value)), "mgs_set_option")
    elif key in DEFAULT_OPTIONS:
        if key == "seg" and int(value) not in (512, 1024, 2048, 4096):
            raise RuntimeError("seg must be 512, 1024, 2048 or 4096")
        DEFAULT_OPTIONS[key] = int(value)
        OPTIONS_VERSION[0] += 1
        push_options()
    else:
        raise RuntimeError(f"unknown option {key}")


def push_options():
    """Hand the per-call defaults to the compiled binding (it fills MgsRasterArgs.opt itself)."""
    from . import _state
    e = _state._EXT[0]
    if e:
        o = DEFAULT_OPTIONS
        e.set_options(o["tight_bins"], o["fast_exp"], o["exact_cull"], o["bin_mode"], o["seg"], o["gm_waves"], o["dbg"],
                      o["table_init"])


def get_option(key: str) -> int:
    if key == "profile":
        return lib().mgs_get_option(key.encode())
    return DEFAULT_OPTIONS[key]


LONG_LIST = 8192  # instances per tile above which 4096-key segments pay (each key is ranked in half as many segments)
VERY_LONG_LIST = 24576  # ... above which the segment sort + rank merge beats the bucket rank (scripts/diag/quick_binmode.py)


def auto_seg(a, opts, mark_R, tiles):
    """The default segment length (2048) is raised to 4096 for shapes whose tiles hold long lists (BASELINE configs[4]'s shape:
    14 000 instances per tile = 7 segments of 2048, every key ranked in the six others; sort + merge 115 us instead of 123,
    profiles/r04_exp_segsort.log): known from the instance count of earlier forwards of the shape.  An explicitly set seg is
    left alone."""
    if opts["seg"] == 2048 and mark_R and mark_R > LONG_LIST * tiles:
        a.opt.seg = 4096
    # ... and the bucket rank (bin_mode 2: a tile's slice in registers, <= 16 384 keys) hands VERY long lists -- more than 24 576
    # instances per tile on average, e.g. 2 000 000 Gaussians on a 128 x 128 image -- to the segment sort + rank merge, whose
    # work is spread over one workgroup per segment (590 against 505 us there; the bucket rank wins at every BASELINE shape)
    if opts["bin_mode"] == 2 and mark_R and mark_R > VERY_LONG_LIST * tiles:
        a.opt.bin_mode = 1
        if opts["seg"] == 2048:
            a.opt.seg = 4096


def fill_options(a, opts=None):
    """Copy per-call options into a.opt (opts: a dict snapshot, default: DEFAULT_OPTIONS); returns the snapshot."""
    o = DEFAULT_OPTIONS if opts is None else opts
    q = a.opt
    q.set = 1
    q.tight_bins, q.fast_exp, q.exact_cull, q.bin_mode = o["tight_bins"], o["fast_exp"], o["exact_cull"], o["bin_mode"]
    q.seg, q.gm_waves, q.dbg, q.table_init = o["seg"], o["gm_waves"], o["dbg"], o["table_init"]
    return o if opts is not None else dict(o)


def profile_read(reset: bool = True):
    """{stage name: (total ms, launches)} from the library's hipEvent stage timers."""
    L = lib()
    n = L.mgs_profile_num_stages()
    ms = (ctypes.c_double * n)()
    cnt = (c_i32 * n)()
    check(L.mgs_profile_read(ms, cnt, int(reset)), "mgs_profile_read")
    return {L.mgs_profile_stage_name(i).decode(): (ms[i], cnt[i]) for i in range(n)}
