"""GPU tests of the tile binning's result contract, stated on the binning's own data: every tile's list (point_list) is that
tile's slice of scattered keys in (depth bits, Gaussian index) order -- what the reference's stable radix sort of
tile << 32 | depth keys yields (RAST/cuda_rasterizer/rasterizer_impl.cu:70-138,306-320).  The depth distributions are the ones
that stress the round-6 bucket rank (mgs_binning.hip: order-preserving buckets in LDS): equal depths, a heavy cluster among
spread-out keys (bitonic fall-back), slices longer than one pass holds, a cluster longer than one pass holds."""
import ctypes

import numpy as np
import pytest
import torch

from manigaussian_amd import _C, _lib
from manigaussian_amd import synthetic as syn

pytestmark = pytest.mark.gpu


def _lists(d, cam, W, H, F, blocking=True):
    """Run a forward through the reference-shaped entry point; return (ranges [T,2], keys_unsorted, point_list).  blocking = False:
    the package-default path (worst-case workspace, nothing waited for: the capacity covers the keys of the direct binning)."""
    dev = torch.device("cuda:0")
    kw = syn.camera_settings_kwargs(cam, 1, F > 0, bg=(0.0, 0.0, 0.0), device=dev)
    e = torch.Tensor([])
    out = _C._forward(kw["bg"], d["means3D"], e, d["language_feature"] if F else e, d["opacities"], d["scales"], d["rotations"],
                      1.0, e, kw["viewmatrix"], kw["projmatrix"], kw["tanfovx"], kw["tanfovy"], H, W, d["shs"], 1,
                      kw["campos"], False, False, F > 0, False, blocking=blocking)
    handle, color, feat, radii, geom, binning, img = out[:7]
    torch.cuda.synchronize()
    if img.numel() == 0:  # the package-default path: ONE allocation [geom | img | binning], handed out as the geometry buffer
        a, base = handle.a, handle.a.geom
        img = geom[a.img - base:a.img - base + a.img_bytes]
        binning = geom[a.binning - base:a.binning - base + a.binning_bytes]
    L = _lib.lib()
    ku, pl, rg = ctypes.c_size_t(), ctypes.c_size_t(), ctypes.c_size_t()
    cap = ctypes.c_int32()
    _lib.check(L.mgs_debug_binning_layout(ctypes.byref(handle.a), 0, ctypes.byref(ku), ctypes.byref(pl), ctypes.byref(rg),
                                          ctypes.byref(cap)), "binning_layout")
    T = ((W + 15) // 16) * ((H + 15) // 16)
    b = binning.cpu().numpy()
    im = img.cpu().numpy()
    n = cap.value
    keys = b[ku.value:ku.value + 8 * n].view(np.uint64)
    plist = b[pl.value:pl.value + 4 * n].view(np.uint32)
    ranges = im[rg.value:rg.value + 8 * T].view(np.uint32).reshape(T, 2)
    # "direct" binning: the forward preprocess wrote the keys itself, tile t's slice at stride * t (no bin scatter launch)
    dk, stride = ctypes.c_size_t(), ctypes.c_int32()
    _lib.check(L.mgs_debug_direct_keys(ctypes.byref(handle.a), 0, ctypes.byref(dk), ctypes.byref(stride)), "direct_keys")
    _lists.direct = stride.value > 0
    if stride.value > 0:
        strided = b[dk.value:dk.value + 8 * T * stride.value].view(np.uint64)
        keys = np.zeros(max(n, int(ranges[-1][1])), np.uint64)
        for t, (lo, hi) in enumerate(ranges):
            keys[int(lo):int(hi)] = strided[t * stride.value:t * stride.value + int(hi) - int(lo)]
    return ranges, keys, plist, int(handle), color


def _check_order(ranges, keys, plist):
    total = 0
    for t, (a, b) in enumerate(ranges):
        a, b = int(a), int(b)
        assert b >= a
        if t:
            assert a == int(ranges[t - 1][1]), "slices are contiguous in tile order"
        want = np.sort(keys[a:b])  # (depth bits << 32 | id): unique keys, so the order is the stable sort's
        got = plist[a:b]
        assert np.array_equal(got, (want & np.uint64(0xffffffff)).astype(np.uint32)), f"tile {t}: {b - a} instances out of order"
        total += b - a
    return total


def _scene(P, seed, dev):
    sc = syn.make_scene(P, F=3, M=4, seed=seed)
    return {k: v.to(dev) for k, v in sc.items()}


def _plane(d, cam, idx, depth, spread):
    """Put Gaussians idx on the plane at view depth `depth` in front of the camera (|offsets| <= spread): close to equal
    depths; exact ties come from duplicated positions below."""
    dev = d["means3D"].device
    wv = cam["world_view_transform"].to(dev)  # transposed world -> view
    c2w = torch.linalg.inv(wv.T)
    right, down, fwd, eye = c2w[:3, 0], c2w[:3, 1], c2w[:3, 2], c2w[:3, 3]
    n = idx.numel()
    g = torch.Generator().manual_seed(7)
    ab = ((torch.rand(n, 2, generator=g) - 0.5) * 2 * spread).to(dev)
    d["means3D"][idx] = eye + depth * fwd + ab[:, :1] * right + ab[:, 1:] * down


CASES = {
    "random_one_tile_3000": dict(P=3000, W=16),
    "random_four_tiles_9000": dict(P=9000, W=32),
    "one_position_5000_ties_broken_by_index": dict(P=5000, W=16, same=5000),
    "cluster_6000_among_2000_bitonic_fallback": dict(P=8000, W=16, same=6000),
    "near_plane_7000": dict(P=7000, W=16, plane=7000),
    "long_slice_40000_several_ranges": dict(P=40000, W=16),
    "long_cluster_20000_among_5000_streamed_rank": dict(P=25000, W=16, same=20000),
    "baseline_shape_100k_128": dict(P=100000, W=128),
    "manigaussian_shape_16384_128": dict(P=16384, W=128),
    # the kernel's size classes, one tile each, either side of every boundary: one key; <= 1 024 keys skip level 1; a round holds
    # 8 192; 16 384 keys fit the registers
    "one_gaussian": dict(P=1, W=16, exact=True),
    "seven_gaussians": dict(P=7, W=16, exact=True),
    "slice_1024_no_level_1": dict(P=1024, W=16, exact=True),
    "slice_1025_level_1": dict(P=1025, W=16, exact=True),
    "slice_8192_one_round": dict(P=8192, W=16, exact=True),
    "slice_8193_two_rounds": dict(P=8193, W=16, exact=True),
    "slice_16384_in_registers": dict(P=16384, W=16, exact=True),
    "slice_16385_streamed": dict(P=16385, W=16, exact=True),
    # the preprocess / scatter partition (csrc/mgs_common.h pre_block): 512 Gaussians per workgroup up to 131 072, 1 024 beyond;
    # a last workgroup holding one Gaussian
    "partition_513_last_workgroup_of_one": dict(P=513, W=32),
    "partition_131072_workgroups_of_512": dict(P=131072, W=64),
    "partition_131073_workgroups_of_1024": dict(P=131073, W=64),
    "ragged_tiles_80x48_30000": dict(P=30000, W=80, H=48),
    "many_tiles_256x256_200000": dict(P=200000, W=256),
}


@pytest.mark.parametrize("name", list(CASES))
@pytest.mark.parametrize("bin_mode", [2, -2, 1], ids=["bucket_rank_keys_by_the_preprocess", "bucket_rank_keys_by_the_scatter_launch",
                                                      "segment_sort_rank_merge"])
def test_every_tile_list_is_its_key_slice_in_depth_then_index_order(name, bin_mode):
    """(bin_mode 2, the default, has the forward preprocess write the keys itself wherever the workspace has room for tiles x P of
    them -- the package-default call of this test: a worst-case workspace; -2 = the same call with MgsOptions.dbg & 32768: the bin
    scatter launch writes them, as for workspaces without that room; the segment sort runs behind a blocking call.)"""
    case = CASES[name]
    direct_off = bin_mode == -2
    bin_mode = abs(bin_mode)
    if bin_mode == 1 and case["P"] > 20000 and case["W"] == 16:
        pytest.skip("the segment sort's one-tile long lists are covered by test_gpu_parity (seg 4096)")
    dev = torch.device("cuda:0")
    P, W, H = case["P"], case["W"], case.get("H", case["W"])
    d = _scene(P, 11, dev)
    cam = syn.circle_cameras(4, W, H, negative_focal=True)[1]
    if case.get("exact"):
        # every Gaussian in front of the camera, small: the one tile's slice holds exactly P keys (tight_bins 0)
        _plane(d, cam, torch.arange(P, device=dev), 1.5, 0.01)
        d["means3D"] += 0.2 * torch.linspace(-1, 1, max(P, 2), device=dev)[:P, None] * torch.linalg.inv(cam["world_view_transform"].to(dev).T)[:3, 2]
        d["scales"][:] = 0.004
    if case.get("same"):
        # duplicated positions in front of the camera: bit-equal depths, bit-equal screen positions
        wv = cam["world_view_transform"].to(dev)
        c2w = torch.linalg.inv(wv.T)
        d["means3D"][: case["same"]] = c2w[:3, 3] + 1.7 * c2w[:3, 2]
        d["scales"][: case["same"]] = 0.01
    if case.get("plane"):
        _plane(d, cam, torch.arange(case["plane"], device=dev), 1.6, 0.02)
    old = {k: _lib.get_option(k) for k in ("bin_mode", "tight_bins", "dbg")}
    try:
        _lib.set_option("bin_mode", bin_mode)
        _lib.set_option("dbg", 32768 if direct_off else 0)
        for tight in (0, 1):
            _lib.set_option("tight_bins", tight)
            ranges, keys, plist, R_ref, color = _lists(d, cam, W, H, 3, blocking=bin_mode != 2)
            n = _check_order(ranges, keys, plist)
            assert n > 0 and torch.isfinite(color).all()
            T = ((W + 15) // 16) * ((H + 15) // 16)
            if bin_mode == 2 and (direct_off or P * T <= (16 << 20)):  # (larger: the worst case exceeds the budget, the call waits for the preprocess on a mark-sized workspace)
                assert _lists.direct == (not direct_off), "which kernel wrote the keys"
            if case.get("exact") and tight == 0:
                assert n == P, f"the one tile's slice should hold exactly {P} keys, holds {n}"
            if tight == 0:
                assert n == R_ref, "tight_bins = 0 bins exactly the reference's 3-sigma-rect instances"
            else:
                assert n <= R_ref
            if case.get("same") and W == 16:
                # the cluster sits in the image and its keys share their depth bits (tight_bins drops the few members whose
                # opacity never reaches 1/255)
                a, b = int(ranges[0][0]), int(ranges[0][1])
                depth_bits = (keys[a:b] >> np.uint64(32)).astype(np.uint32)
                assert np.bincount(np.unique(depth_bits, return_inverse=True)[1]).max() >= 0.99 * case["same"]
    finally:
        for k, v in old.items():
            _lib.set_option(k, v)
