"""Oracle A -- pure-PyTorch CPU restatement of the reference rasterizer whose BACKWARD comes from
autograd, not from a restatement of backward.cu.  It is the independent pin for the hand-derived
gradients of Oracle B (oracle/mgs_oracle.c) and of the HIP kernels, and it is the
"CPU PyTorch reference rasterizer" of BASELINE.json configs[0].

TEST INFRASTRUCTURE ONLY -- imported by tests/ and bench.py's cpu_baseline leg, never by the
product path.  Pinned against outputs of the reference's own kernels on the small cases of
tests/golden/ref (tests/test_oracle.py::test_oracle_a_matches_reference_kernels).

RAST = third_party/gaussian-splatting/submodules/diff-gaussian-rasterization (reference tree).

Forward semantics follow RAST/cuda_rasterizer/forward.cu:156-257 (preprocess) and :262-398
(render), binning follows rasterizer_impl.cu:70-138,280-320.  Two places where the reference's
analytic backward is NOT the derivative of its forward are reproduced with stop-gradients so
that autograd yields what backward.cu yields:
  Q1  alpha = min(0.99, o*G): backward.cu:513-590 propagates through o*G even when the min
      saturates  ->  straight-through min.
  Q2  frustum clamp t.x = clamp(t.x/t.z)*t.z: backward.cu:175-176,262-264 zeroes d/dt.x when
      clamped and treats the clamped t.x as a constant w.r.t. t.z  ->  detach the clamped value.
(The 1e-7 in denom2inv, backward.cu:203, is a <=1.3e-5 relative deviation and is not modelled.)
"""
import math

import torch

SH_C0 = 0.28209479177387814
SH_C1 = 0.4886025119029199
SH_C2 = [1.0925484305920792, -1.0925484305920792, 0.31539156525252005, -1.0925484305920792,
         0.5462742152960396]
SH_C3 = [-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154,
         -0.4570457994644658, 1.445305721320277, -0.5900435899266435]
BLOCK = 16


def _sh_to_rgb(deg, sh, dirs):
    """forward.cu:21-72.  sh [P,M,3], dirs [P,3] (normalised)."""
    res = SH_C0 * sh[:, 0]
    if deg > 0:
        x, y, z = dirs[:, 0:1], dirs[:, 1:2], dirs[:, 2:3]
        res = res - SH_C1 * y * sh[:, 1] + SH_C1 * z * sh[:, 2] - SH_C1 * x * sh[:, 3]
        if deg > 1:
            xx, yy, zz, xy, yz, xz = x * x, y * y, z * z, x * y, y * z, x * z
            res = (res + SH_C2[0] * xy * sh[:, 4] + SH_C2[1] * yz * sh[:, 5]
                   + SH_C2[2] * (2.0 * zz - xx - yy) * sh[:, 6] + SH_C2[3] * xz * sh[:, 7]
                   + SH_C2[4] * (xx - yy) * sh[:, 8])
            if deg > 2:
                res = (res + SH_C3[0] * y * (3.0 * xx - yy) * sh[:, 9] + SH_C3[1] * xy * z * sh[:, 10]
                       + SH_C3[2] * y * (4.0 * zz - xx - yy) * sh[:, 11]
                       + SH_C3[3] * z * (2.0 * zz - 3.0 * xx - 3.0 * yy) * sh[:, 12]
                       + SH_C3[4] * x * (4.0 * zz - xx - yy) * sh[:, 13] + SH_C3[5] * z * (xx - yy) * sh[:, 14]
                       + SH_C3[6] * x * (xx - 3.0 * yy) * sh[:, 15])
    res = res + 0.5
    return torch.clamp_min(res, 0.0), (res < 0)


def _clamp_q2(v, lim, tz):
    """t.x = min(lim, max(-lim, v)) * t.z with the reference-backward semantics (Q2)."""
    lim_t = torch.full_like(v, lim)
    c = torch.minimum(lim_t, torch.maximum(-lim_t, v))
    clamped = (v < -lim) | (v > lim)
    val = c * tz
    return torch.where(clamped, val.detach(), val)


def rasterize(means3D, opacities, settings, shs=None, colors_precomp=None, language_feature=None,
              scales=None, rotations=None, cov3D_precomp=None, means2D=None, dtype=None):
    """Differentiable restatement.  Returns (color [3,H,W], feature [F,H,W] or [1], radii [P] int32, aux).

    means2D, if given ([P,3], zeros), is the reference's screen-space gradient holder
    (gaussian_renderer/__init__.py:28): its autograd gradient equals dL_dmean2D of backward.cu:581-582
    (NDC units, 0.5*W / 0.5*H folded in)."""
    dt = dtype or means3D.dtype
    cast = lambda t: None if t is None else t.to(dt)
    means3D, opacities = cast(means3D), cast(opacities)
    shs, colors_precomp, language_feature = cast(shs), cast(colors_precomp), cast(language_feature)
    scales, rotations, cov3D_precomp = cast(scales), cast(rotations), cast(cov3D_precomp)
    P = means3D.shape[0]
    H, W = int(settings.image_height), int(settings.image_width)
    V = settings.viewmatrix.detach().to(dt).reshape(4, 4).cpu()
    PM = settings.projmatrix.detach().to(dt).reshape(4, 4).cpu()
    campos = settings.campos.detach().to(dt).cpu()
    bg = settings.bg.detach().to(dt).cpu()
    tanx, tany = float(settings.tanfovx), float(settings.tanfovy)
    inc = bool(settings.include_feature)
    F = language_feature.shape[1] if (inc and language_feature is not None and language_feature.numel()) else 0
    gx, gy = (W + BLOCK - 1) // BLOCK, (H + BLOCK - 1) // BLOCK
    color0 = bg.reshape(3, 1, 1).expand(3, H, W)
    if P == 0:  # rasterize_points.cu:92: outputs stay zero-filled
        return (torch.zeros(3, H, W, dtype=dt), torch.zeros(F, H, W, dtype=dt) if inc else torch.zeros(1, dtype=dt),
                torch.zeros(0, dtype=torch.int32), {})
    focal_x, focal_y = W / (2.0 * tanx), H / (2.0 * tany)

    ones = torch.ones(P, 1, dtype=dt)
    hom = torch.cat([means3D, ones], 1)
    p_view = hom @ V[:, :3]                      # auxiliary.h:58-66
    p_hom = hom @ PM                             # auxiliary.h:68-77
    p_w = 1.0 / (p_hom[:, 3] + 0.0000001)
    ndc = p_hom[:, :2] * p_w[:, None]
    in_front = p_view[:, 2] > 0.2                # auxiliary.h:154

    if cov3D_precomp is not None and cov3D_precomp.numel():
        c6 = cov3D_precomp
        Sigma = torch.stack([c6[:, 0], c6[:, 1], c6[:, 2], c6[:, 1], c6[:, 3], c6[:, 4], c6[:, 2], c6[:, 4],
                             c6[:, 5]], 1).reshape(P, 3, 3)
    else:                                        # forward.cu:119-153, quaternion NOT normalised
        s = float(settings.scale_modifier) * scales
        r, x, y, z = rotations[:, 0], rotations[:, 1], rotations[:, 2], rotations[:, 3]
        Rstd = torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y),
                            2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x),
                            2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], 1).reshape(P, 3, 3)
        RS = Rstd * s[:, None, :]
        Sigma = RS @ RS.transpose(1, 2)

    # forward.cu:75-114
    tz = p_view[:, 2]
    tx = _clamp_q2(p_view[:, 0] / tz, 1.3 * tanx, tz)
    ty = _clamp_q2(p_view[:, 1] / tz, 1.3 * tany, tz)
    zero = torch.zeros_like(tz)
    Jstd = torch.stack([focal_x / tz, zero, -(focal_x * tx) / (tz * tz), zero, focal_y / tz,
                        -(focal_y * ty) / (tz * tz), zero, zero, zero], 1).reshape(P, 3, 3)
    Rw2c = V[:3, :3].t()                         # viewmatrix is the transposed world->view
    A = Jstd @ Rw2c
    cov2 = A @ Sigma @ A.transpose(1, 2)
    ca, cb, cc = cov2[:, 0, 0] + 0.3, cov2[:, 0, 1], cov2[:, 1, 1] + 0.3
    det = ca * cc - cb * cb
    ok = in_front & (det != 0)
    det_safe = torch.where(ok, det, torch.ones_like(det))
    conic = torch.stack([cc / det_safe, -cb / det_safe, ca / det_safe], 1)
    with torch.no_grad():
        mid = 0.5 * (ca + cc)
        disc = torch.sqrt(torch.clamp_min(mid * mid - det, 0.1))
        radius = torch.ceil(3.0 * torch.sqrt(torch.maximum(mid + disc, mid - disc)))
        radius = torch.where(ok, radius, torch.zeros_like(radius))
    pix = torch.stack([((ndc[:, 0] + 1.0) * W - 1.0) * 0.5, ((ndc[:, 1] + 1.0) * H - 1.0) * 0.5], 1)
    if means2D is not None:
        pix = pix + means2D[:, :2].to(dt) * torch.tensor([0.5 * W, 0.5 * H], dtype=dt)

    with torch.no_grad():                        # auxiliary.h:46-56 (trunc toward zero, clamp to grid)
        pd, rd = pix.detach(), radius
        rx0 = ((pd[:, 0] - rd) / BLOCK).trunc().clamp(0, gx).long()
        ry0 = ((pd[:, 1] - rd) / BLOCK).trunc().clamp(0, gy).long()
        rx1 = ((pd[:, 0] + rd + BLOCK - 1) / BLOCK).trunc().clamp(0, gx).long()
        ry1 = ((pd[:, 1] + rd + BLOCK - 1) / BLOCK).trunc().clamp(0, gy).long()
        touched = (rx1 - rx0) * (ry1 - ry0)
        touched = torch.where(ok, touched, torch.zeros_like(touched))
        vis = touched > 0
        radii = torch.where(vis, radius, torch.zeros_like(radius)).to(torch.int32)

    if shs is not None and shs.numel():
        d = means3D - campos[None]
        d = d / d.norm(dim=1, keepdim=True)
        rgb, clamped = _sh_to_rgb(int(settings.sh_degree), shs, d)
    else:
        rgb, clamped = colors_precomp, None

    # binning: rasterizer_impl.cu:70-111 keys, :306-311 stable sort, :116-138 ranges
    with torch.no_grad():
        depth_bits = p_view[:, 2].detach().float().contiguous().view(torch.int32).long()
        ids, tiles = [], []
        mw = int((rx1 - rx0)[vis].max()) if vis.any() else 0
        mh = int((ry1 - ry0)[vis].max()) if vis.any() else 0
        for dy in range(mh):
            for dx in range(mw):
                m = vis & (rx0 + dx < rx1) & (ry0 + dy < ry1)
                i = m.nonzero()[:, 0]
                ids.append(i)
                tiles.append((ry0[i] + dy) * gx + rx0[i] + dx)
        if ids:
            ids, tiles = torch.cat(ids), torch.cat(tiles)
        else:
            ids, tiles = torch.zeros(0, dtype=torch.long), torch.zeros(0, dtype=torch.long)
        o1 = torch.sort(ids, stable=True)[1]      # emit order = Gaussian index order (then y, x)
        ids, tiles = ids[o1], tiles[o1]
        key = (tiles << 32) | depth_bits[ids]
        o2 = torch.sort(key, stable=True)[1]
        point_list, tile_sorted = ids[o2], tiles[o2]
        R = int(point_list.numel())
        counts = torch.bincount(tile_sorted, minlength=gx * gy) if R else torch.zeros(gx * gy, dtype=torch.long)
        ends = torch.cumsum(counts, 0)
        starts = ends - counts

    out_f = torch.zeros(F, H, W, dtype=dt) if inc else torch.zeros(1, dtype=dt)
    final_T = torch.ones(H, W, dtype=dt)
    n_contrib = torch.zeros(H, W, dtype=torch.int32)
    color_tiles, feat_tiles = {}, {}
    for tile in range(gx * gy):
        ty_, tx_ = tile // gx, tile % gx
        y0, x0 = ty_ * BLOCK, tx_ * BLOCK
        y1, x1 = min(y0 + BLOCK, H), min(x0 + BLOCK, W)
        s0, e0 = int(starts[tile]), int(ends[tile])
        if e0 == s0:
            continue
        L = point_list[s0:e0]
        yy, xx = torch.meshgrid(torch.arange(y0, y1), torch.arange(x0, x1), indexing="ij")
        pxf = xx.reshape(-1, 1).to(dt)
        pyf = yy.reshape(-1, 1).to(dt)
        dx = pix[L, 0][None, :] - pxf
        dy = pix[L, 1][None, :] - pyf
        cn = conic[L]
        power = -0.5 * (cn[:, 0][None] * dx * dx + cn[:, 2][None] * dy * dy) - cn[:, 1][None] * dx * dy
        G = torch.exp(torch.clamp_max(power, 0.0))
        a_raw = opacities[L, 0][None] * G
        alpha = a_raw + (torch.clamp_max(a_raw, 0.99) - a_raw).detach()          # Q1
        valid = (power <= 0) & (alpha >= 1.0 / 255.0)                            # forward.cu:345-352
        one_m = torch.where(valid, 1.0 - alpha, torch.ones_like(alpha))
        T_after = torch.cumprod(one_m, 1)
        T_before = torch.cat([torch.ones_like(T_after[:, :1]), T_after[:, :-1]], 1)
        term = valid & (T_before * (1.0 - alpha) < 0.0001)                       # forward.cu:353-360
        keep = valid & (torch.cumsum(term.to(torch.int32), 1) == 0)
        w = torch.where(keep, alpha * T_before, torch.zeros_like(alpha))
        Tf = torch.cumprod(torch.where(keep, 1.0 - alpha, torch.ones_like(alpha)), 1)[:, -1]
        C = w @ rgb[L] + Tf[:, None] * bg[None]                                  # forward.cu:388
        hh, ww = y1 - y0, x1 - x0
        color_tiles[tile] = (y0, y1, x0, x1, C.t().reshape(3, hh, ww))
        if inc:
            feat_tiles[tile] = (w @ language_feature[L]).t().reshape(F, hh, ww)  # no bg, forward.cu:393
        with torch.no_grad():
            final_T[y0:y1, x0:x1] = Tf.reshape(hh, ww)
            idx1 = torch.arange(1, e0 - s0 + 1)[None].expand_as(keep)
            n_contrib[y0:y1, x0:x1] = torch.where(keep, idx1, torch.zeros_like(idx1)).max(1)[0].reshape(hh, ww).int()
    # assemble without in-place writes on a graph leaf
    rows = []
    frows = []
    for ty_ in range(gy):
        row, frow = [], []
        for tx_ in range(gx):
            tile = ty_ * gx + tx_
            y0, x0 = ty_ * BLOCK, tx_ * BLOCK
            y1, x1 = min(y0 + BLOCK, H), min(x0 + BLOCK, W)
            if tile in color_tiles:
                row.append(color_tiles[tile][4])
            else:
                row.append(bg.reshape(3, 1, 1).expand(3, y1 - y0, x1 - x0))
            if inc:
                frow.append(feat_tiles.get(tile, torch.zeros(F, y1 - y0, x1 - x0, dtype=dt)))
        rows.append(torch.cat(row, 2))
        if inc:
            frows.append(torch.cat(frow, 2))
    out_c = torch.cat(rows, 1)
    if inc:
        out_f = torch.cat(frows, 1)
    aux = dict(num_rendered=R, point_list=point_list, ranges=torch.stack([starts, ends], 1), final_T=final_T,
               n_contrib=n_contrib, means2D=pix.detach(), conic=conic.detach(), depths=p_view[:, 2].detach(),
               rgb=None if rgb is None else rgb.detach(), clamped=clamped, radii=radii)
    return out_c, out_f, radii, aux


def forward_backward(inputs: dict, settings, d_color, d_feat=None, dtype=torch.float32):
    """Run fwd + autograd bwd.  inputs: means3D, opacities, [shs|colors_precomp], [language_feature],
    scales, rotations | cov3D_precomp.  Returns (color, feat, radii, grads, aux); grads keyed like
    oracle_b.backward (means2D in the reference's NDC units)."""
    leaves = {}
    for k, v in inputs.items():
        if v is None:
            continue
        leaves[k] = v.detach().clone().to(dtype).requires_grad_(True)
    P = leaves["means3D"].shape[0]
    leaves["means2D"] = torch.zeros(P, 3, dtype=dtype, requires_grad=True)
    color, feat, radii, aux = rasterize(
        leaves["means3D"], leaves["opacities"], settings, shs=leaves.get("shs"),
        colors_precomp=leaves.get("colors_precomp"), language_feature=leaves.get("language_feature"),
        scales=leaves.get("scales"), rotations=leaves.get("rotations"), cov3D_precomp=leaves.get("cov3D_precomp"),
        means2D=leaves["means2D"], dtype=dtype)
    loss = (color * d_color.to(dtype)).sum()
    if settings.include_feature and d_feat is not None and feat.numel() > 1:
        loss = loss + (feat * d_feat.to(dtype)).sum()
    names = [k for k in leaves]
    gs = torch.autograd.grad(loss, [leaves[k] for k in names], allow_unused=True) if loss.requires_grad else [None] * len(names)
    grads = {}
    for k, g in zip(names, gs):
        grads[k] = torch.zeros_like(leaves[k]) if g is None else g
    if "shs" in grads:
        grads["sh"] = grads.pop("shs")
    return color.detach(), feat.detach(), radii, grads, aux
