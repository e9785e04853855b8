"""render(): the per-view Python prologue ManiGaussian runs before the rasterizer.

Mirror of agents/manigaussian_bc/gaussian_renderer/__init__.py:17-94 (same signature, same returned dict,
same choices: sh_degree = 3 unless SH features are given, features L2-normalised with a 1e-12 guard, a
zeros [N,3] placeholder when no language features are given).  The reference file itself is executed
unmodified against this repository's `diff_gaussian_rasterization` package on the GPU by
tests/test_integration.py (from a build-time byte copy that is never committed).
"""
import math

import torch

from .rasterizer import GaussianRasterizationSettings, GaussianRasterizer


def render(data, idx, pts_xyz, rotations, scales, opacity, bg_color, pts_rgb=None, features_color=None,
           features_language=None):
    device = pts_xyz.device
    bg = torch.tensor(bg_color, dtype=torch.float32, device=device)
    # gradient holder for the 2D means (gaussian_renderer/__init__.py:28-32)
    screenspace_points = torch.zeros_like(pts_xyz, dtype=torch.float32, requires_grad=True, device=device) + 0
    try:
        screenspace_points.retain_grad()
    except Exception:
        pass
    view = data["novel_view"]
    if "tanfov_host" in view:  # manigaussian_amd.camera.TargetCache: host copies, no device read-back per view
        tanfovx, tanfovy = view["tanfov_host"][idx]
        height, width = view["size_host"][idx]
    else:                      # the reference's dict: four scalar device reads (gaussian_renderer/__init__.py:35-39)
        tanfovx, tanfovy = math.tan(view["FovX"][idx] * 0.5), math.tan(view["FovY"][idx] * 0.5)
        height, width = int(view["height"][idx]), int(view["width"][idx])
    settings = GaussianRasterizationSettings(
        image_height=height, image_width=width, tanfovx=tanfovx, tanfovy=tanfovy,
        bg=bg, scale_modifier=1.0, viewmatrix=view["world_view_transform"][idx],
        projmatrix=view["full_proj_transform"][idx], sh_degree=3 if features_color is None else 1,
        campos=view["camera_center"][idx], prefiltered=False, debug=False,
        include_feature=(features_language is not None))
    rasterizer = GaussianRasterizer(raster_settings=settings)
    shs = colors_precomp = None
    if features_color is not None:
        shs = features_color
    else:
        assert pts_rgb is not None
        colors_precomp = pts_rgb
    if features_language is not None:
        feats = features_language / (features_language.norm(dim=-1, keepdim=True) + 1e-12)
    else:
        feats = torch.zeros((opacity.shape[0], 3), dtype=opacity.dtype, device=opacity.device)
    image, feature_image, radii = rasterizer(means3D=pts_xyz, means2D=screenspace_points, shs=shs,
                                             colors_precomp=colors_precomp, language_feature_precomp=feats,
                                             opacities=opacity, scales=scales, rotations=rotations,
                                             cov3D_precomp=None)
    return {"render": image, "render_embed": feature_image, "viewspace_points": screenspace_points, "radii": radii}
