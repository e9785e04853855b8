"""Shared helpers for the parity tests: run the same seeded scene through Oracle B (CPU) and the HIP path."""
import types

import numpy as np
import torch

from manigaussian_amd import synthetic as syn


def scene_case(P, F=3, M=4, W=128, H=128, neg=True, colors_precomp=False, bg=(0.1, 0.2, 0.3), sh_degree=1,
               include_feature=True, seed=0, cam_index=1, unnormalized_rot=False, cov3d=False, scale_modifier=1.0):
    sc = syn.make_scene(P, F=F if include_feature else 0, M=M, seed=seed, colors_precomp=colors_precomp,
                        unnormalized_rot=unnormalized_rot)
    if cov3d:  # precomputed covariance instead of scale/rotation
        R = sc.pop("rotations")
        s = sc.pop("scales")
        r, x, y, z = R[:, 0], R[:, 1], R[:, 2], R[:, 3]
        Rm = torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y),
                          2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x),
                          2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], 1).reshape(-1, 3, 3)
        RS = Rm * s[:, None, :]
        Sg = RS @ RS.transpose(1, 2)
        sc["cov3D_precomp"] = torch.stack([Sg[:, 0, 0], Sg[:, 0, 1], Sg[:, 0, 2], Sg[:, 1, 1], Sg[:, 1, 2],
                                           Sg[:, 2, 2]], 1).contiguous()
    cam = syn.circle_cameras(4, W, H, negative_focal=neg)[cam_index]
    kw = syn.camera_settings_kwargs(cam, sh_degree, include_feature, bg=bg)
    kw["scale_modifier"] = float(scale_modifier)  # enters computeCov3D (forward.cu:122-126) and the cov3D backward, whose
    dC, dF = syn.make_cotangents(W, H, F if include_feature else 0)  # dL_dscale omits it (backward.cu:295,325-327)
    return sc, cam, kw, dC, dF


def run_oracle_b(sc, kw, dC, dF):
    from oracle import oracle_b
    st = types.SimpleNamespace(**kw)
    color, feat, radii, state = oracle_b.forward(
        sc["means3D"], sc["opacities"], st, shs=sc.get("shs"), colors_precomp=sc.get("colors_precomp"),
        language_feature=sc.get("language_feature"), scales=sc.get("scales"), rotations=sc.get("rotations"),
        cov3D_precomp=sc.get("cov3D_precomp"))
    grads = oracle_b.backward(state, dC, dF)
    return color, feat, radii, grads, state


def stored_inputs(z, sc):
    """The inputs a golden file was generated from (bit-exact), after checking that today's generator still produces
    them (derived arrays such as cov3D_precomp go through a matmul whose rounding depends on the host CPU)."""
    out = {}
    for k, v in sc.items():
        a = z[f"in_{k}"]
        assert np.allclose(v.numpy(), a, rtol=1e-5, atol=1e-9), f"scene generator drifted: {k}"
        out[k] = torch.from_numpy(a)
    return out


def run_reference(sc, kw, dC, dF):
    """The reference's own kernels (oracle/_ref via oracle/ref_cuda.py; GPU, F = 3 only).  Returns
    (color, feat, radii, grads by Oracle-B name, num_rendered)."""
    from oracle import ref_cuda
    st = types.SimpleNamespace(**kw)
    return ref_cuda.forward_backward(
        sc["means3D"], sc["opacities"], st, dC, dF, shs=sc.get("shs"), colors_precomp=sc.get("colors_precomp"),
        language_feature=sc.get("language_feature"), scales=sc.get("scales"), rotations=sc.get("rotations"),
        cov3D_precomp=sc.get("cov3D_precomp"))


def run_hip(sc, cam, dC, dF, sh_degree, include_feature, bg, device="cuda:0", debug=False, scale_modifier=1.0):
    from manigaussian_amd import GaussianRasterizationSettings, GaussianRasterizer
    dev = torch.device(device)
    kw = syn.camera_settings_kwargs(cam, sh_degree, include_feature, bg=bg, device=dev, debug=debug)
    kw["scale_modifier"] = float(scale_modifier)
    leaves = {k: v.to(dev).requires_grad_(True) for k, v in sc.items()}
    P = sc["means3D"].shape[0]
    means2D = torch.zeros(P, 3, device=dev, requires_grad=True)
    rast = GaussianRasterizer(GaussianRasterizationSettings(**kw))
    color, feat, radii = rast(means3D=leaves["means3D"], means2D=means2D, opacities=leaves["opacities"],
                              shs=leaves.get("shs"), colors_precomp=leaves.get("colors_precomp"),
                              language_feature_precomp=leaves.get("language_feature"), scales=leaves.get("scales"),
                              rotations=leaves.get("rotations"), cov3D_precomp=leaves.get("cov3D_precomp"))
    loss = (color * dC.to(dev)).sum()
    if include_feature:
        loss = loss + (feat * dF.to(dev)).sum()
    loss.backward()
    torch.cuda.synchronize()
    grads = {k: v.grad.detach().cpu() for k, v in leaves.items() if v.grad is not None}
    grads["means2D"] = means2D.grad.detach().cpu()
    return color.detach().cpu(), feat.detach().cpu(), radii.cpu(), grads


# name in the HIP leaves -> name in the oracle's gradient dict
GRAD_KEYS = {"means3D": "means3D", "means2D": "means2D", "opacities": "opacities", "scales": "scales",
             "rotations": "rotations", "shs": "sh", "colors_precomp": "colors_precomp",
             "language_feature": "language_feature", "cov3D_precomp": "cov3D"}


def grad_errors(g_hip, g_ref):
    """{name: (max abs err, max |ref|)}"""
    out = {}
    for k, v in g_hip.items():
        r = g_ref[GRAD_KEYS[k]].reshape(v.shape)
        out[k] = ((v - r).abs().max().item() if v.numel() else 0.0, r.abs().max().item() if r.numel() else 0.0)
    return out


# ---- tolerances -----------------------------------------------------------------------------------------------------
# Contract (BASELINE.json north_star): 1e-4 on images, 1e-3 * max|g| on gradients.  The algorithm has three hard per-pair
# decisions (alpha < 1/255 -> skip, power > 0 -> skip, T(1-alpha) < 1e-4 -> stop); a pair that sits within float rounding of
# one of them may legitimately be decided differently by two correct implementations, and a flipped pair moves a pixel by
# alpha*T*|c| <= |c|/255 ~ 4e-3 (alpha threshold) or ~1e-4*|c| (T threshold).
#  * against ORACLE B (gcc on the host: another exp(), other FMA contraction) such flips are routine: the pixels / Gaussians
#    Oracle B itself marks as sitting within 2e-5 (relative) of a threshold get FRAGILE_TOL, everything else the contract.
#  * against the REFERENCE'S OWN KERNELS on the same GPU two things differ: v_exp_f32 vs ocml expf (2e-7 relative), and the
#    ASSOCIATION of the transmittance product -- the reference multiplies (1 - alpha) pair by pair, the chunk-parallel forward
#    multiplies per-chunk products into the transmittance entering a chunk (then walks the chunk sequentially): the same
#    real number, another last bit, and T (1 - alpha) < 1e-4 is a hard decision.  Round 4 measured both
#    (test_live_reference_exact_exp_*): at 100 000 Gaussians / 128^2 no pixel differs by more than 9.5e-7 under EITHER exp;
#    at 500 000 / 256^2 (~1e9 pairs) ONE pixel sat at 1.16e-4 / 2.03e-4 -- the same pixel, the same value, under v_exp_f32
#    AND under ocml expf, with the exact sequential transmittance chain, with and without the culls: none of the render's
#    decisions was the cause.  It was the PREPROCESS: cov3D a few ulps from the reference's on 90 % of the Gaussians (another
#    fused-multiply-add pattern for the same algebra), conics up to 85 ulps behind it.  With the roundings pinned
#    (test_preprocess_is_bit_identical_to_the_reference_kernels) NO pixel of any live-reference case exceeds the 1e-4
#    contract (worst: 8.9e-5, one pixel of the 500 000 case; <= 1e-6 at 100 000).  Bounds: unmarked pixels 2e-5, a marked
#    pixel REF_FRAGILE_TOL = 5e-4 (view batches: an alpha < 1/255 flip moves a pixel by up to T |c| / 255), NO pixel of a
#    single-view image above the 1e-4 contract; gradients meet the contract (1e-3 of the max) everywhere, marked or not.
FRAGILE_TOL = 1e-2
FRAGILE_MAX_FRACTION = 0.02
REF_FRAGILE_TOL = 5e-4
REF_FRAGILE_GRAD_TOL = 1e-3
REF_MAX_PIXELS_ABOVE_CONTRACT = 0


def report(tag, **stats):
    """Append measured parity statistics to $MGS_PARITY_REPORT (one JSON object per line), so that one GPU run documents how
    far inside the bounds the implementation sits (profiles/r03_parity_report.jsonl)."""
    import json
    import os
    path = os.environ.get("MGS_PARITY_REPORT")
    if path:
        with open(path, "a") as f:
            f.write(json.dumps({"tag": str(tag)[:160], **{k: (float(v) if isinstance(v, (float, np.floating)) else v)
                                                         for k, v in stats.items()}}) + "\n")


def image_errors(img_hip, img_ref, state, rel_eps=2e-5):
    """(max err over robust pixels, max err over threshold-fragile pixels, fragile fraction).

    Fragile pixels (oracle_b.fragile_mask) sit within rel_eps of one of the reference's hard decisions
    (alpha < 1/255, T(1-alpha) < 1e-4, power > 0); there a 1-ulp exp() difference legitimately changes which
    Gaussians are blended, in the CUDA reference as much as here, so they get FRAGILE_TOL instead of 1e-4."""
    from oracle import oracle_b
    frag = oracle_b.fragile_mask(state, rel_eps)
    err = (img_hip - img_ref).abs()
    if err.dim() == 3:
        err = err.max(0)[0]
    robust = err[~frag].max().item() if (~frag).any() else 0.0
    fragile = err[frag].max().item() if frag.any() else 0.0
    return robust, fragile, frag.float().mean().item()


FRAGILE_GRAD_TOL = 1e-2


def grad_errors_split(g_hip, g_ref, state, rel_eps=2e-5):
    """{name: (robust err, fragile err, max |ref|)}: errors split by oracle_b.fragile_gaussians -- Gaussians with
    a (pixel, Gaussian) pair within rel_eps of a hard threshold, where a 1-ulp difference flips a whole pair in
    or out of the blend (the same holds between the CUDA reference and any other exp())."""
    from oracle import oracle_b
    fg = oracle_b.fragile_gaussians(state, rel_eps)
    out = {}
    for k, v in g_hip.items():
        r = g_ref[GRAD_KEYS[k]].reshape(v.shape)
        if v.numel() == 0:
            out[k] = (0.0, 0.0, 0.0)
            continue
        d = (v - r).abs().reshape(v.shape[0], -1).max(1)[0]
        out[k] = (d[~fg].max().item() if (~fg).any() else 0.0, d[fg].max().item() if fg.any() else 0.0,
                  r.abs().max().item())
    return out, fg.float().mean().item()
