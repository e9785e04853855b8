"""Synthetic Gaussian scenes and cameras for parity tests and bench.py.

Scene statistics follow what ManiGaussian's Gaussian regressor emits
(agents/manigaussian_bc/models_embed.py:245-253, conf/method/ManiGaussian_BC.yaml:139-140,
conf/config.yaml:21 scene bounds); cameras are processed exactly like
NeuralRenderer.get_novel_calib (agents/manigaussian_bc/neural_rendering.py:205-248) with
getProjectionMatrix / getWorld2View2 / focal2fov restated from
agents/manigaussian_bc/graphics_utils.py:17-53.

Everything is generated on the CPU with fixed seeds so that the CPU oracle and the GPU path
see bit-identical inputs.
"""
import math
from typing import Dict, List, Optional

import numpy as np
import torch

SCENE_BOUNDS = (-0.3, -0.5, 0.6, 0.7, 0.5, 1.6)  # conf/config.yaml:21


def make_scene(P: int, F: int = 3, M: int = 4, seed: int = 0, colors_precomp: bool = False,
               unnormalized_rot: bool = False) -> Dict[str, torch.Tensor]:
    """Seeded synthetic Gaussian set (CPU float32 tensors)."""
    g = torch.Generator().manual_seed(seed)
    lo = torch.tensor(SCENE_BOUNDS[:3])
    hi = torch.tensor(SCENE_BOUNDS[3:])
    means3D = lo + (hi - lo) * torch.rand(P, 3, generator=g)
    scales = torch.clamp_max(torch.exp(math.log(0.02) + 0.5 * torch.randn(P, 3, generator=g)), 0.05)
    rot = torch.randn(P, 4, generator=g)
    if not unnormalized_rot:
        rot = torch.nn.functional.normalize(rot, dim=-1)
    opacities = torch.sigmoid(-2.0 + torch.randn(P, 1, generator=g))
    out = dict(means3D=means3D, scales=scales, rotations=rot, opacities=opacities)
    if colors_precomp:
        out["colors_precomp"] = torch.rand(P, 3, generator=g)
    else:
        out["shs"] = 0.3 * torch.randn(P, M, 3, generator=g)
    if F > 0:
        out["language_feature"] = torch.nn.functional.normalize(torch.randn(P, F, generator=g), dim=-1)
    return {k: v.float().contiguous() for k, v in out.items()}


def make_cotangents(W: int, H: int, F: int, seed: int = 1):
    g = torch.Generator().manual_seed(seed)
    d_color = torch.randn(3, H, W, generator=g)
    d_feat = torch.randn(F, H, W, generator=g) if F > 0 else None
    return d_color, d_feat


# --- restated from agents/manigaussian_bc/graphics_utils.py:17-53 -----------------------------

def focal2fov(focal: float, pixels: float) -> float:
    return 2 * math.atan(pixels / (2 * focal))


def get_world2view2(R, t, translate=np.array([0.0, 0.0, 0.0]), scale=1.0):
    Rt = np.zeros((4, 4))
    Rt[:3, :3] = R.transpose()
    Rt[:3, 3] = t
    Rt[3, 3] = 1.0
    C2W = np.linalg.inv(Rt)
    cam_center = (C2W[:3, 3] + translate) * scale
    C2W[:3, 3] = cam_center
    return np.float32(np.linalg.inv(C2W))


def get_projection_matrix(znear, zfar, K, h, w):
    near_fx = znear / K[0, 0]
    near_fy = znear / K[1, 1]
    left = -(w - K[0, 2]) * near_fx
    right = K[0, 2] * near_fx
    bottom = (K[1, 2] - h) * near_fy
    top = K[1, 2] * near_fy
    Pm = torch.zeros(4, 4)
    Pm[0, 0] = 2.0 * znear / (right - left)
    Pm[1, 1] = 2.0 * znear / (top - bottom)
    Pm[0, 2] = (right + left) / (right - left)
    Pm[1, 2] = (top + bottom) / (top - bottom)
    Pm[3, 2] = 1.0
    Pm[2, 2] = zfar / (zfar - znear)
    Pm[2, 3] = -(zfar * znear) / (zfar - znear)
    return Pm


def novel_calib(c2w: np.ndarray, K: np.ndarray, W: int, H: int, znear: float = 0.1, zfar: float = 4.0):
    """One camera through NeuralRenderer.get_novel_calib (neural_rendering.py:217-236)."""
    extr = np.linalg.inv(c2w)
    R = np.array(extr[:3, :3], np.float32).reshape(3, 3).transpose(1, 0)
    T = np.array(extr[:3, 3], np.float32)
    FovX = focal2fov(K[0, 0], W)
    FovY = focal2fov(K[1, 1], H)
    projection_matrix = get_projection_matrix(znear, zfar, K, H, W).transpose(0, 1)
    world_view_transform = torch.tensor(get_world2view2(R, T)).transpose(0, 1)
    full_proj_transform = world_view_transform.unsqueeze(0).bmm(projection_matrix.unsqueeze(0)).squeeze(0)
    camera_center = world_view_transform.inverse()[3, :3]
    return dict(FovX=float(FovX), FovY=float(FovY), width=W, height=H,
                world_view_transform=world_view_transform.contiguous(),
                full_proj_transform=full_proj_transform.contiguous(),
                camera_center=camera_center.contiguous())


def look_at_c2w(eye, target, up=(0.0, 0.0, 1.0), flip_xy: bool = False) -> np.ndarray:
    """cam2world with the camera looking down +z, x right, y down (OpenCV).  flip_xy negates the
    x and y axes: the PyRep vision-sensor frame that goes with its negative focal length
    (third_party/PyRep/pyrep/objects/vision_sensor.py:188)."""
    eye = np.asarray(eye, np.float64)
    fwd = np.asarray(target, np.float64) - eye
    fwd /= np.linalg.norm(fwd)
    right = np.cross(fwd, np.asarray(up, np.float64))
    right /= np.linalg.norm(right)
    down = np.cross(fwd, right)
    if flip_xy:
        right, down = -right, -down
    c2w = np.eye(4)
    c2w[:3, 0], c2w[:3, 1], c2w[:3, 2], c2w[:3, 3] = right, down, fwd, eye
    return c2w


def circle_cameras(V: int, W: int = 128, H: int = 128, negative_focal: bool = True,
                   fov_deg: float = 40.0, phase: float = 0.0) -> List[dict]:
    """V look-at cameras on a circle: eye = target + (1.3 cos t, 1.3 sin t, 0.9).
    negative_focal=True is the production regime."""
    target = np.array([0.2, 0.0, 0.9])
    f = (W / 2) / math.tan(math.radians(fov_deg / 2))
    cams = []
    for k in range(V):
        th = phase + 2 * math.pi * k / max(V, 1)
        eye = target + np.array([1.3 * math.cos(th), 1.3 * math.sin(th), 0.9])
        c2w = look_at_c2w(eye, target, flip_xy=negative_focal)
        fs = -f if negative_focal else f
        K = np.array([[fs, 0, W / 2], [0, fs, H / 2], [0, 0, 1]], np.float64)
        cams.append(novel_calib(c2w, K, W, H))
    return cams


def camera_settings_kwargs(cam: dict, sh_degree: int, include_feature: bool,
                           bg=(0.0, 0.0, 0.0), device="cpu", debug: bool = False) -> dict:
    """Field values for GaussianRasterizationSettings, built as render() does
    (agents/manigaussian_bc/gaussian_renderer/__init__.py:35-52)."""
    return dict(
        image_height=int(cam["height"]), image_width=int(cam["width"]),
        tanfovx=math.tan(cam["FovX"] * 0.5), tanfovy=math.tan(cam["FovY"] * 0.5),
        bg=torch.tensor(bg, dtype=torch.float32, device=device), scale_modifier=1.0,
        viewmatrix=cam["world_view_transform"].to(device), projmatrix=cam["full_proj_transform"].to(device),
        sh_degree=sh_degree, campos=cam["camera_center"].to(device), prefiltered=False, debug=debug,
        include_feature=include_feature)
