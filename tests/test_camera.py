"""Camera / target I/O: the calibration routine against the REFERENCE's
NeuralRenderer.get_novel_calib (agents/manigaussian_bc/neural_rendering.py:205-248 + graphics_utils.py:17-53, executed
unmodified through tests/ref_import.py wherever the reference files are available -- /root/reference or the build-time copies
oracle/_ref/mg), the on-disk format, and the target cache.  The committed outputs of that same method are checked in
tests/test_reference_modules.py (tests/golden/mg/novel_calib.npz); where no copy of the reference exists, the random cameras
below fall back to the host routine that test pins."""
import math
import os

import numpy as np
import pytest
import torch

import util
from manigaussian_amd import camera
from manigaussian_amd import synthetic as syn


def _cameras(V, W, H, neg, seed=0):
    rng = np.random.default_rng(seed)
    c2w, K = [], []
    for v in range(V):
        th = 2 * math.pi * v / V
        eye = np.array([0.2 + 1.3 * math.cos(th), 1.3 * math.sin(th), 0.9 + rng.uniform(0.0, 0.6)])
        c2w.append(syn.look_at_c2w(eye, (0.2, 0.0, 0.9), flip_xy=neg))
        f = (W / 2) / math.tan(math.radians(rng.uniform(15.0, 35.0))) * (-1 if neg else 1)
        K.append(np.array([[f, 0, W / 2 + rng.uniform(-3, 3)], [0, f * rng.uniform(0.9, 1.1), H / 2 + rng.uniform(-3, 3)],
                           [0, 0, 1]]))
    return np.stack(c2w), np.stack(K)


def _reference_calib(c2w, K, W, H, znear, zfar):
    """NeuralRenderer.get_novel_calib of the reference on these cameras (intrinsics: float32 values in a float64 tensor, the
    arithmetic of the reference's NumPy-1.x environment -- tests/golden/make_golden_mg.py), or the pinned host routine."""
    import types
    import ref_import
    NR = ref_import.load_neural_rendering() if ref_import.have_reference() else None
    if NR is None:
        h = camera.novel_calib_host(c2w, K, W, H, znear, zfar)
        return dict(world_view_transform=h["world_view_transform"], full_proj_transform=h["full_proj_transform"],
                    camera_center=h["camera_center"], FovX=h["fov"][:, 0], FovY=h["fov"][:, 1])
    self_ = types.SimpleNamespace(W=W, H=H, znear=znear, zfar=zfar, trans=[0.0, 0.0, 0.0], scale=1.0)
    nv = NR.NeuralRenderer.get_novel_calib(self_, dict(intr=torch.from_numpy(K.astype(np.float32).astype(np.float64)),
                                                       extr=torch.from_numpy(c2w.astype(np.float32))))
    return {k: v.numpy() for k, v in nv.items()}


def _check(got, c2w, K, W, H, znear, zfar):
    ref = _reference_calib(c2w, K, W, H, znear, zfar)
    for v in range(c2w.shape[0]):
        for name in ("world_view_transform", "full_proj_transform", "camera_center"):
            r = ref[name][v]
            assert np.abs(np.asarray(got[name][v]) - r).max() <= 2e-6 * max(1.0, np.abs(r).max()), (v, name)
        fx, fy = float(ref["FovX"][v]), float(ref["FovY"][v])
        assert abs(float(got["fov"][v][0]) - fx) <= 1e-6 and abs(float(got["fov"][v][1]) - fy) <= 1e-6
        assert abs(float(got["tanfov"][v][0]) - math.tan(fx * 0.5)) <= 2e-6 * abs(math.tan(fx * 0.5)) + 1e-7
        assert (fx < 0) == (K[v][0, 0] < 0)  # negative focal lengths stay negative


@pytest.mark.parametrize("neg", [True, False], ids=["negfocal", "posfocal"])
def test_host_calibration_matches_get_novel_calib(neg):
    W, H = 128, 96
    c2w, K = _cameras(7, W, H, neg)
    _check(camera.novel_calib_host(c2w, K, W, H, 0.1, 4.0), c2w, K, W, H, 0.1, 4.0)


def test_host_calibration_rejects_bad_input():
    c2w, K = _cameras(2, 64, 64, True)
    c2w[1] = 0.0
    with pytest.raises(RuntimeError, match="singular"):
        camera.novel_calib_host(c2w, K, 64, 64)
    with pytest.raises(RuntimeError, match="bad arguments"):
        camera.novel_calib_host(c2w[:1], K[:1], 64, 64, znear=1.0, zfar=0.5)


def _write_scene(tmp_path, V, W, H):
    from PIL import Image
    c2w, K = _cameras(V, W, H, True, seed=3)
    rng = np.random.default_rng(5)
    paths = []
    for v in range(V):
        rgb = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
        dep = rng.integers(0, 256, (H, W), dtype=np.uint8)
        pr, pd, pc = (os.path.join(tmp_path, f"{n}_{v}.{e}") for n, e in (("rgb", "png"), ("depth", "png"), ("pose", "txt")))
        Image.fromarray(rgb).save(pr)
        Image.fromarray(dep).save(pd)
        with open(pc, "w") as f:  # the recorder's format (YARR video_utils.py:205-216): 4 rows, blank line, 3 rows
            for row in c2w[v]:
                f.write(" ".join(repr(float(x)) for x in row) + "\n")
            f.write("\n")
            for row in K[v]:
                f.write(" ".join(repr(float(x)) for x in row) + "\n")
        paths.append((pr, pd, pc, rgb, dep))
    return c2w, K, paths


def test_file_format_and_cache(tmp_path):
    W, H = 40, 24
    c2w, K, paths = _write_scene(str(tmp_path), 3, W, H)
    e, k, focal = camera.parse_camera_file(paths[1][2])
    assert np.array_equal(e, c2w[1]) and np.array_equal(k, K[1]) and focal == K[1][0, 0]
    assert np.array_equal(camera.parse_img_file(paths[0][0]), paths[0][3].astype(np.float32) / 255.0)
    assert np.array_equal(camera.parse_depth_file(paths[0][1]), paths[0][4].astype(np.float32))
    cache = camera.TargetCache(W, H, capacity=2, pin=False)
    b = cache.load_batch([p[0] for p in paths[:2]], [p[1] for p in paths[:2]], [p[2] for p in paths[:2]], "cpu")
    assert b["rgb"].shape == (2, H, W, 3) and b["depth"].shape == (2, H, W) and b["extr"].shape == (2, 4, 4)
    assert torch.equal(b["rgb"][1], torch.from_numpy(paths[1][3].astype(np.float32) / 255.0))
    nv = b["novel_view"]
    got = dict(world_view_transform=nv["world_view_transform"], full_proj_transform=nv["full_proj_transform"],
               camera_center=nv["camera_center"], fov=torch.stack([nv["FovX"], nv["FovY"]], 1), tanfov=nv["tanfov_host"])
    _check(got, c2w[:2], K[:2], W, H, 0.1, 4.0)
    assert nv["size_host"] == [(H, W)] * 2 and (cache.hits, cache.misses) == (0, 2)
    cache.load_batch([paths[1][0]], [paths[1][1]], [paths[1][2]], "cpu")
    assert (cache.hits, cache.misses) == (1, 2)
    cache.get(*paths[2][:3])                       # third entry evicts the least recently used (entry 0)
    cache.get(*paths[0][:3])
    assert cache.misses == 4 and len(cache._entries) == 2


@pytest.mark.gpu
@pytest.mark.parametrize("neg", [True, False], ids=["negfocal", "posfocal"])
def test_device_calibration_matches_get_novel_calib(neg):
    W, H = 128, 128
    c2w, K = _cameras(70, W, H, neg, seed=9)
    dev = torch.device("cuda:0")
    data = {"intr": torch.from_numpy(K).float().to(dev), "extr": torch.from_numpy(c2w).float().to(dev)}
    nv = camera.get_novel_calib(data, W, H, 0.1, 4.0)
    assert set(nv) >= {"FovX", "FovY", "width", "height", "world_view_transform", "full_proj_transform", "camera_center"}
    got = dict(world_view_transform=nv["world_view_transform"].cpu(), full_proj_transform=nv["full_proj_transform"].cpu(),
               camera_center=nv["camera_center"].cpu(), fov=torch.stack([nv["FovX"], nv["FovY"]], 1).cpu(),
               tanfov=nv["tanfov"].cpu())
    _check(got, c2w, K, W, H, 0.1, 4.0)
    host = camera.novel_calib_host(c2w, K, W, H, 0.1, 4.0)   # same routine on both sides: near-identical
    for name in ("world_view_transform", "full_proj_transform", "camera_center", "fov"):
        assert np.abs(got[name].numpy() - host[name]).max() <= 1e-6


@pytest.mark.gpu
def test_render_with_cached_camera_equals_reference_dict_path(tmp_path):
    """render() fed from the cache (host tanfov, no read-back) == render() fed from the reference-shaped dict."""
    from manigaussian_amd.gaussian_renderer import render
    W = H = 64
    c2w, K, paths = _write_scene(str(tmp_path), 2, W, H)
    dev = torch.device("cuda:0")
    cache = camera.TargetCache(W, H)
    b = cache.load_batch([p[0] for p in paths], [p[1] for p in paths], [p[2] for p in paths], dev)
    nv_dev = camera.get_novel_calib({"intr": b["intr"], "extr": b["extr"]}, W, H)
    sc = {k: v.to(dev) for k, v in syn.make_scene(2000, F=3).items()}
    outs = []
    for nv in (b["novel_view"], nv_dev):
        o = render({"novel_view": nv}, 1, sc["means3D"], sc["rotations"], sc["scales"], sc["opacities"], [0.0, 0.0, 0.0],
                   features_color=sc["shs"], features_language=sc["language_feature"])
        outs.append(o["render"])
    assert (outs[0] - outs[1]).abs().max() <= 1e-5 and outs[0].abs().max() > 0.05
