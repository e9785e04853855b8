"""Camera / target I/O off the critical path.

What the reference does every training step (agents/manigaussian_bc/qattention_manigaussian_bc_agent.py:716-739 and
neural_rendering.py:205-248): PIL-decodes two RGB + two depth PNGs and parses two camera text files synchronously inside
`update()`, uploads them, then `get_novel_calib` copies the intrinsics/extrinsics BACK to the host (`.cpu().numpy()`, a
device synchronisation), inverts with numpy and uploads seven small tensors again; `render()` finally reads FovX/FovY/
width/height back once more (`math.tan(data['novel_view']['FovX'][idx] * 0.5)`, gaussian_renderer/__init__.py:35-36).

Here:
  * `parse_camera_file` / `parse_img_file` / `parse_depth_file` read the same on-disk format
    (nerf_data/<t>/{images,depths,poses}/<i>.{png,png,txt}; camera txt = 4x4 cam2world, blank line, 3x3 K --
    qattention_manigaussian_bc_agent.py:86-129, written by third_party/YARR/yarr/utils/video_utils.py:205-262);
  * `TargetCache` decodes each (rgb, depth, camera) triple ONCE into pinned host memory together with its calibration
    (`mgs_novel_calib_host`: world_view_transform, full_proj_transform, camera_center, Fov and tan(Fov/2) as host floats),
    and `load_batch` turns a batch of paths into device tensors with asynchronous copies only;
  * `get_novel_calib` is the drop-in for NeuralRenderer.get_novel_calib when intrinsics/extrinsics are already device
    tensors: one kernel launch (`mgs_novel_calib`), no host round trip; same keys as the reference's dict plus `tanfov`.
`manigaussian_amd.gaussian_renderer.render` uses `novel_view['tanfov_host']` / `['size_host']` when present, which removes the
last per-view synchronisation.
"""
import collections
import ctypes
import math

import numpy as np
import torch

from . import _lib


def parse_camera_file(file_path):
    """-> (cam2world [4,4] float64, K [3,3] float64, focal).  Format of qattention_manigaussian_bc_agent.py:86-113: four
    rows of the extrinsic, one separator line, three rows of the intrinsic."""
    with open(file_path, "r") as f:
        lines = f.readlines()
    extr = np.array([float(y) for x in lines[0:4] for y in x.split()], dtype=np.float64).reshape(4, 4)
    intr = np.array([float(y) for x in lines[5:8] for y in x.split()], dtype=np.float64).reshape(3, 3)
    return extr, intr, intr[0, 0]


def parse_img_file(file_path):
    """RGB image in [0, 1], float32 [H, W, 3] (qattention_manigaussian_bc_agent.py:115-121)."""
    from PIL import Image
    return np.asarray(Image.open(file_path).convert("RGB")).astype(np.float32) / 255.0


def parse_depth_file(file_path):
    """8-bit depth image as float32 [H, W], not normalised (qattention_manigaussian_bc_agent.py:123-129)."""
    from PIL import Image
    return np.asarray(Image.open(file_path).convert("L")).astype(np.float32)


def _fp(a):
    return ctypes.c_void_p(a.ctypes.data)


def novel_calib_host(extr, intr, W, H, znear=0.1, zfar=4.0, trans=(0.0, 0.0, 0.0), scale=1.0):
    """Host calibration of V cameras.  extr [V,4,4] (cam2world), intr [V,3,3] -> dict of float32 numpy arrays:
    world_view_transform [V,4,4], full_proj_transform [V,4,4], camera_center [V,3], fov [V,2], tanfov [V,2]."""
    L = _lib.lib()
    e = np.ascontiguousarray(np.asarray(extr, dtype=np.float32).reshape(-1, 16))
    k = np.ascontiguousarray(np.asarray(intr, dtype=np.float32).reshape(-1, 9))
    V = e.shape[0]
    if k.shape[0] != V:
        raise ValueError("extr and intr disagree on the number of cameras")
    out = dict(world_view_transform=np.empty((V, 4, 4), np.float32), full_proj_transform=np.empty((V, 4, 4), np.float32),
               camera_center=np.empty((V, 3), np.float32), fov=np.empty((V, 2), np.float32),
               tanfov=np.empty((V, 2), np.float32))
    _lib.check(L.mgs_novel_calib_host(V, _fp(e), _fp(k), int(W), int(H), float(znear), float(zfar), float(trans[0]),
                                      float(trans[1]), float(trans[2]), float(scale), _fp(out["world_view_transform"]),
                                      _fp(out["full_proj_transform"]), _fp(out["camera_center"]), _fp(out["fov"]),
                                      _fp(out["tanfov"])), "novel_calib_host")
    return out


def get_novel_calib(data, W, H, znear=0.1, zfar=4.0, trans=(0.0, 0.0, 0.0), scale=1.0):
    """Drop-in for NeuralRenderer.get_novel_calib (neural_rendering.py:205-248): data['intr'] [bs,3,3], data['extr']
    [bs,4,4] (cam2world) on a HIP device -> the same dict (FovX, FovY, width, height, world_view_transform,
    full_proj_transform, camera_center), computed by one kernel without leaving the device, plus 'tanfov' [bs,2]."""
    L = _lib.lib()
    intr, extr = data["intr"], data["extr"]
    if not intr.is_cuda:
        raise RuntimeError("get_novel_calib needs tensors on a HIP device (novel_calib_host is the host-side routine)")
    dev = intr.device
    bs = intr.shape[0]
    k = intr.float().contiguous()
    e = extr.float().contiguous()
    o = dict(dtype=torch.float32, device=dev)
    wvt, fpt = torch.empty((bs, 4, 4), **o), torch.empty((bs, 4, 4), **o)
    centre, fov, tanfov = torch.empty((bs, 3), **o), torch.empty((bs, 2), **o), torch.empty((bs, 2), **o)
    with torch.cuda.device(dev):
        stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        _lib.check(L.mgs_novel_calib(bs, e.data_ptr(), k.data_ptr(), int(W), int(H), float(znear), float(zfar),
                                     float(trans[0]), float(trans[1]), float(trans[2]), float(scale), wvt.data_ptr(),
                                     fpt.data_ptr(), centre.data_ptr(), fov.data_ptr(), tanfov.data_ptr(), None, stream),
                   "novel_calib")
    return {"FovX": fov[:, 0], "FovY": fov[:, 1], "width": torch.full((bs,), int(W), device=dev),
            "height": torch.full((bs,), int(H), device=dev), "world_view_transform": wvt, "full_proj_transform": fpt,
            "camera_center": centre, "tanfov": tanfov}


class TargetCache:
    """Decoded supervision targets + calibrated cameras, keyed by their file paths, kept in pinned host memory (LRU).

    entry = dict(rgb [H,W,3] f32, depth [H,W] f32, extr [4,4] f32, intr [3,3] f32, world_view_transform, full_proj_transform,
    camera_center, fov [2] (all pinned CPU tensors) and tanfov = (tanfovx, tanfovy) as Python floats)."""

    def __init__(self, W, H, znear=0.1, zfar=4.0, trans=(0.0, 0.0, 0.0), scale=1.0, capacity=8192, pin=None):
        self.W, self.H, self.znear, self.zfar, self.trans, self.scale = W, H, znear, zfar, trans, scale
        self.capacity = capacity
        self.pin = torch.cuda.is_available() if pin is None else pin
        self._entries = collections.OrderedDict()
        self.hits = self.misses = 0

    def _host(self, a):
        t = torch.from_numpy(np.ascontiguousarray(a))
        return t.pin_memory() if self.pin else t

    def get(self, rgb_path, depth_path, camera_path):
        key = (rgb_path, depth_path, camera_path)
        e = self._entries.get(key)
        if e is not None:
            self._entries.move_to_end(key)
            self.hits += 1
            return e
        self.misses += 1
        extr, intr, _ = parse_camera_file(camera_path)
        c = novel_calib_host(extr[None], intr[None], self.W, self.H, self.znear, self.zfar, self.trans, self.scale)
        e = dict(rgb=self._host(parse_img_file(rgb_path)), depth=self._host(parse_depth_file(depth_path)),
                 extr=self._host(extr.astype(np.float32)), intr=self._host(intr.astype(np.float32)),
                 world_view_transform=self._host(c["world_view_transform"][0]),
                 full_proj_transform=self._host(c["full_proj_transform"][0]),
                 camera_center=self._host(c["camera_center"][0]), fov=self._host(c["fov"][0]),
                 tanfov=(float(c["tanfov"][0, 0]), float(c["tanfov"][0, 1])))
        self._entries[key] = e
        while len(self._entries) > self.capacity:
            self._entries.popitem(last=False)
        return e

    def load_batch(self, rgb_paths, depth_paths, camera_paths, device):
        """What qattention_manigaussian_bc_agent.py:716-739 + get_novel_calib produce for a batch of paths: device tensors
        rgb [bs,H,W,3], depth [bs,H,W], extr [bs,4,4], intr [bs,3,3] and the novel_view dict -- by asynchronous copies from
        pinned memory, no decode and no synchronisation on a warm cache."""
        es = [self.get(r, d, c) for r, d, c in zip(rgb_paths, depth_paths, camera_paths)]
        bs = len(es)

        def up(name):
            t = torch.stack([e[name] for e in es])  # staging buffer; pinned so that the upload is asynchronous
            return (t.pin_memory() if self.pin else t).to(device, non_blocking=True)
        fov = up("fov")
        novel = {"FovX": fov[:, 0], "FovY": fov[:, 1], "width": torch.full((bs,), int(self.W), device=device),
                 "height": torch.full((bs,), int(self.H), device=device),
                 "world_view_transform": up("world_view_transform"), "full_proj_transform": up("full_proj_transform"),
                 "camera_center": up("camera_center"), "tanfov_host": [e["tanfov"] for e in es],
                 "size_host": [(int(self.H), int(self.W))] * bs}
        return dict(rgb=up("rgb"), depth=up("depth"), extr=up("extr"), intr=up("intr"), novel_view=novel)


def focal2fov(focal, pixels):
    """MG/graphics_utils.py:51-52 (negative focal lengths give negative FoV; kept)."""
    return 2 * math.atan(pixels / (2 * focal))
