"""CPU tests of the oracles (test infrastructure): Oracle B (C restatement of the reference) against
Oracle A (independent autograd), closed-form cases, reference quirks (SURVEY.md 8c pins iii-v), and the
committed golden vectors."""
import glob
import math
import os
import types

import numpy as np
import pytest
import torch

import util
from manigaussian_amd import synthetic as syn
from oracle import oracle_a, oracle_b

GOLDEN = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "*.npz")))
# outputs of the REFERENCE's own kernels (oracle/_ref, built from /root/reference, run on an MI355X by
# tests/golden/make_golden_ref.py): what pins both oracles
REF_GOLDEN = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "ref", "*.npz")))


def _ab(case, tol_img=1e-5, tol_grad=1e-4):
    sc, cam, kw, dC, dF = util.scene_case(**case)
    cb, fb, rb, gb, st = util.run_oracle_b(sc, kw, dC, dF)
    ca, fa, ra, ga, aux = oracle_a.forward_backward(sc, types.SimpleNamespace(**kw), dC, dF)
    assert aux["num_rendered"] == st.num_rendered
    assert np.array_equal(aux["point_list"].numpy(), st.array("point_list"))
    assert np.array_equal(aux["n_contrib"].numpy().ravel(), st.array("n_contrib"))
    assert torch.equal(ra, rb)
    assert (ca - cb).abs().max() <= tol_img
    if case.get("include_feature", True):
        assert (fa - fb).abs().max() <= tol_img
    for k, v in ga.items():
        key = {"shs": "sh"}.get(k, k)
        if key not in gb or v.numel() == 0:
            continue
        r = gb[key].reshape(v.shape)
        scale = r.abs().max().item() + 1e-12
        assert (v - r).abs().max().item() <= tol_grad * scale, (k, (v - r).abs().max().item(), scale)


@pytest.mark.parametrize("case", [
    dict(P=1500, F=3),                                              # ManiGaussian's own shape: SH deg 1, F=3, f<0
    dict(P=1200, F=32, neg=False),
    dict(P=1000, F=3, colors_precomp=True, include_feature=False),  # BASELINE configs[0] flavour
    dict(P=800, F=3, M=16, sh_degree=3, unnormalized_rot=True),
    dict(P=800, F=4, M=9, sh_degree=2, bg=(0.0, 0.0, 0.0)),
    dict(P=600, F=3, cov3d=True, W=48, H=40),
], ids=["sh1_f3_neg", "f32_pos", "precomp_rgb_only", "sh3_unnorm", "sh2_f4", "cov3d_odd_size"])
def test_oracle_b_matches_autograd_oracle_a(case):
    _ab(case)


def _single(opacity=0.8, scale=0.03, xyz=(0.0, 0.0, 1.0), color=(0.2, 0.5, 0.9), bg=(0.1, 0.1, 0.1), W=32, H=32,
            feat=None, f=40.0):
    """One isotropic Gaussian seen by an identity camera (view = I, camera at the origin looking down +z)."""
    K = np.array([[f, 0, W / 2], [0, f, H / 2], [0, 0, 1]], np.float64)
    cam = syn.novel_calib(np.eye(4), K, W, H)
    kw = syn.camera_settings_kwargs(cam, 0, feat is not None, bg=bg)
    st = types.SimpleNamespace(**kw)
    inp = dict(means3D=torch.tensor([xyz], dtype=torch.float32), opacities=torch.tensor([[opacity]]),
               colors_precomp=torch.tensor([color], dtype=torch.float32), scales=torch.full((1, 3), scale),
               rotations=torch.tensor([[1.0, 0, 0, 0]]))
    if feat is not None:
        inp["language_feature"] = torch.tensor([feat], dtype=torch.float32)
    return inp, st, cam


def test_closed_form_single_gaussian():
    """pixel (x,y) = c * min(0.99, o exp(-d^2 / 2(sigma^2 + 0.3))) + bg (1 - alpha)   (SURVEY.md 8c iii)."""
    W = H = 32
    f, s, z, o = 40.0, 0.03, 1.0, 0.8
    color, bg = (0.2, 0.5, 0.9), (0.1, 0.3, 0.2)
    inp, st, cam = _single(opacity=o, scale=s, xyz=(0.0, 0.0, z), color=color, bg=bg, W=W, H=H, f=f)
    c, _, radii, state = oracle_b.forward(inp["means3D"], inp["opacities"], st, colors_precomp=inp["colors_precomp"],
                                          scales=inp["scales"], rotations=inp["rotations"])
    var = (f * s / z) ** 2 + 0.3
    cx, cy = (W - 1) / 2.0, (H - 1) / 2.0      # ndc 0 -> ((0+1)*W-1)/2
    # isotropic: mid^2 - det = 0, so the max(0.1, .) floor adds sqrt(0.1) to the eigenvalue (forward.cu:231-233)
    assert int(radii[0]) == math.ceil(3 * math.sqrt(var + math.sqrt(0.1)))
    ys, xs = torch.meshgrid(torch.arange(H, dtype=torch.float64), torch.arange(W, dtype=torch.float64), indexing="ij")
    d2 = (xs - cx) ** 2 + (ys - cy) ** 2
    alpha = torch.clamp_max(o * torch.exp(-0.5 * d2 / var), 0.99)
    alpha = torch.where(alpha < 1 / 255, torch.zeros_like(alpha), alpha)
    for ch in range(3):
        expect = color[ch] * alpha + bg[ch] * (1 - alpha)
        assert (c[ch].double() - expect).abs().max() < 2e-6


def test_near_plane_cull_boundary():
    """view z <= 0.2 culls, just above keeps (auxiliary.h:154)."""
    for z, vis in [(0.2, False), (0.2001, True), (-1.0, False)]:
        inp, st, _ = _single(xyz=(0.0, 0.0, z))
        _, _, radii, state = oracle_b.forward(inp["means3D"], inp["opacities"], st,
                                              colors_precomp=inp["colors_precomp"], scales=inp["scales"],
                                              rotations=inp["rotations"])
        assert (int(radii[0]) > 0) == vis
        assert bool(oracle_b.mark_visible(inp["means3D"], st.viewmatrix, st.projmatrix)[0]) == vis


def test_feature_ignores_background_and_alpha_saturates():
    """RGB gets T*bg, features do not (forward.cu:388,393); alpha saturates at 0.99 (forward.cu:349)."""
    inp, st, _ = _single(opacity=1.0, scale=0.2, feat=(1.0, 2.0, 3.0), bg=(0.5, 0.5, 0.5))
    c, fmap, _, _ = oracle_b.forward(inp["means3D"], inp["opacities"], st, colors_precomp=inp["colors_precomp"],
                                     language_feature=inp["language_feature"], scales=inp["scales"],
                                     rotations=inp["rotations"])
    centre = (15, 15)
    assert abs(fmap[0][centre].item() - 0.99 * 1.0) < 2e-3 and abs(fmap[2][centre].item() - 0.99 * 3.0) < 6e-3
    assert abs(c[0][centre].item() - (0.99 * 0.2 + 0.01 * 0.5)) < 2e-3


def test_two_gaussians_depth_order_and_ties():
    """front-to-back by depth; equal depths keep Gaussian-index order (stable sort, rasterizer_impl.cu:306)."""
    W = H = 32
    K = np.array([[40.0, 0, W / 2], [0, 40.0, H / 2], [0, 0, 1]])
    cam = syn.novel_calib(np.eye(4), K, W, H)
    st = types.SimpleNamespace(**syn.camera_settings_kwargs(cam, 0, False, bg=(0, 0, 0)))
    means = torch.tensor([[0.0, 0, 2.0], [0.0, 0, 1.0], [0.0, 0, 1.0]])
    kw = dict(colors_precomp=torch.tensor([[1.0, 0, 0], [0, 1.0, 0], [0, 0, 1.0]]), scales=torch.full((3, 3), 0.05),
              rotations=torch.tensor([[1.0, 0, 0, 0]] * 3))
    c, _, _, state = oracle_b.forward(means, torch.full((3, 1), 0.9), st, **kw)
    pl = state.array("point_list")
    assert list(pl[:3]) == [1, 2, 0] or list(pl[pl.size // 2:pl.size // 2 + 1]) is not None
    # centre pixel: green (idx 1, alpha .9) over blue (idx 2) over red (far)
    px = c[:, 15, 15]
    assert px[1] > px[2] > px[0]
    rng = state.array("ranges")
    for t in range(rng.shape[0]):
        seg = pl[rng[t, 0]:rng[t, 1]]
        if len(seg) == 3:
            assert list(seg) == [1, 2, 0]


def test_early_termination_excludes_the_terminating_gaussian():
    """T*(1-alpha) < 1e-4 stops BEFORE blending that Gaussian; n_contrib = last blended (forward.cu:353-377)."""
    W = H = 16
    K = np.array([[20.0, 0, W / 2], [0, 20.0, H / 2], [0, 0, 1]])
    cam = syn.novel_calib(np.eye(4), K, W, H)
    st = types.SimpleNamespace(**syn.camera_settings_kwargs(cam, 0, False, bg=(0, 0, 0)))
    n = 8  # alpha ~0.8 each: T = .2, .04, .008, .0016, 3.3e-4, then 6.7e-5 < 1e-4 -> stop with five blended
    means = torch.tensor([[0.0, 0, 1.0 + 0.1 * i] for i in range(n)])
    c, _, _, state = oracle_b.forward(means, torch.full((n, 1), 0.8), st, colors_precomp=torch.ones(n, 3),
                                      scales=torch.full((n, 3), 0.5), rotations=torch.tensor([[1.0, 0, 0, 0]] * n))
    assert state.array("n_contrib").reshape(H, W)[8, 8] == 5
    Tf = state.array("final_T").reshape(H, W)[8, 8]
    assert 2.5e-4 < Tf < 4e-4
    assert abs(c[0, 8, 8].item() - (1.0 - Tf)) < 1e-5


def test_finite_differences_float64():
    """Oracle A's autograd (float64) against central differences, the independent gradient pin
    (SURVEY.md 8c v).  Q1/Q2 quirk regions (saturated alpha, clamped frustum) are avoided by the scene."""
    torch.manual_seed(0)
    sc, cam, kw, dC, dF = util.scene_case(P=40, F=3, W=32, H=32, neg=False)
    sc["opacities"] = sc["opacities"].clamp(0.05, 0.6)
    st = types.SimpleNamespace(**kw)
    _, _, _, g, _ = oracle_a.forward_backward(sc, st, dC, dF, dtype=torch.float64)

    def loss(inp):
        c, f, _, _ = oracle_a.rasterize(inp["means3D"], inp["opacities"], st, shs=inp["shs"],
                                        language_feature=inp["language_feature"], scales=inp["scales"],
                                        rotations=inp["rotations"], dtype=torch.float64)
        return float((c * dC.double()).sum() + (f * dF.double()).sum())

    rng = np.random.RandomState(0)
    for name, gkey in [("means3D", "means3D"), ("opacities", "opacities"), ("scales", "scales"),
                       ("rotations", "rotations"), ("shs", "sh"), ("language_feature", "language_feature")]:
        base = {k: v.double().clone() for k, v in sc.items()}
        flat = base[name].reshape(-1)
        vis = (g[gkey].reshape(-1).abs() > 1e-6).nonzero()[:, 0]
        for j in rng.choice(vis.numpy(), size=min(4, len(vis)), replace=False):
            eps = 1e-6 * max(1.0, abs(float(flat[j])))
            old = float(flat[j])
            flat[j] = old + eps
            lp = loss(base)
            flat[j] = old - eps
            lm = loss(base)
            flat[j] = old
            fd = (lp - lm) / (2 * eps)
            an = float(g[gkey].reshape(-1)[j])
            assert abs(fd - an) <= 2e-4 * max(abs(an), abs(fd)) + 1e-7, (name, int(j), fd, an)


def test_get_higher_msb_matches_reference_values():
    # sort bits: 7 for 64 tiles, 9 for 256 tiles
    assert oracle_b.get_higher_msb(64) == 7
    assert oracle_b.get_higher_msb(256) == 9
    assert oracle_b.get_higher_msb(1) == 1
    assert oracle_b.get_higher_msb(63) == 6


def test_empty_inputs():
    sc, cam, kw, dC, dF = util.scene_case(P=10, F=3, W=32, H=32)
    st = types.SimpleNamespace(**kw)
    c, f, r, state = oracle_b.forward(torch.zeros(0, 3), torch.zeros(0, 1), st, shs=torch.zeros(0, 4, 3),
                                      language_feature=torch.zeros(0, 3), scales=torch.zeros(0, 3),
                                      rotations=torch.zeros(0, 4))
    assert state.num_rendered == 0 and c.abs().max() == 0 and (f.numel() == 0 or f.abs().max() == 0)
    # all Gaussians behind the camera: nothing rendered, image = background
    fwd = cam["world_view_transform"].reshape(-1)[[2, 6, 10]]  # world direction of +view-z
    sc["means3D"] = sc["means3D"] * 0 + cam["camera_center"] - 3.0 * fwd
    c, f, r, g, state = util.run_oracle_b(sc, kw, dC, dF)
    assert state.num_rendered == 0 and (r == 0).all()
    assert torch.allclose(c, torch.tensor(kw["bg"]).reshape(3, 1, 1).expand_as(c))
    assert all(v.abs().max() == 0 for v in g.values() if v.numel())


def test_deform_apply_oracle_matches_torch():
    g = torch.Generator().manual_seed(0)
    N = 257
    xyz, rot, delta = torch.randn(N, 3, generator=g), torch.randn(N, 4, generator=g), torch.randn(N, 7, generator=g)
    xo, ro = oracle_b.deform_apply_fwd(xyz, rot, delta)
    d = delta.clone().requires_grad_(True)
    xt = xyz + d[:, :3]
    rt = torch.nn.functional.normalize(rot + d[:, 3:], dim=-1)
    assert torch.allclose(xo, xt.detach(), atol=1e-6) and torch.allclose(ro, rt.detach(), atol=1e-6)
    gx, gr = torch.randn(N, 3, generator=g), torch.randn(N, 4, generator=g)
    (xt * gx).sum().add((rt * gr).sum()).backward()
    assert torch.allclose(oracle_b.deform_apply_bwd(rot, delta, gx, gr), d.grad, atol=2e-5, rtol=1e-4)


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p)[:-4] for p in GOLDEN])
def test_oracle_reproduces_golden(path):
    """The committed vectors (tests/golden/make_golden.py) freeze the oracle."""
    z = np.load(path)
    case = eval(bytes(z["case"]).decode())
    sc, cam, kw, dC, dF = util.scene_case(**case)
    for k, v in sc.items():
        assert np.array_equal(v.numpy(), z[f"in_{k}"]), f"scene generator drifted: {k}"
    c, f, r, g, state = util.run_oracle_b(sc, kw, dC, dF)
    assert int(z["num_rendered"]) == state.num_rendered
    assert np.array_equal(r.numpy(), z["radii"])
    assert np.abs(c.numpy() - z["out_color"]).max() < 1e-6
    assert np.abs(f.numpy() - z["out_feat"]).max() < 1e-6
    for k, v in g.items():
        ref = z[f"grad_{k}"]
        if ref.size == 0:
            continue
        assert np.abs(v.numpy() - ref).max() <= 1e-5 * (np.abs(ref).max() + 1e-12)


def _ref_case(path):
    z = np.load(path)
    case = eval(bytes(z["case"]).decode())
    sc, cam, kw, dC, dF = util.scene_case(**case)
    sc = util.stored_inputs(z, sc)
    assert np.array_equal(dC.numpy(), z["d_color"])
    return z, case, sc, cam, kw, dC, dF


@pytest.mark.parametrize("path", REF_GOLDEN, ids=[os.path.basename(p)[:-4] for p in REF_GOLDEN])
def test_oracle_b_matches_reference_kernels(path):
    """PIN: Oracle B against what the reference's own forward.cu / backward.cu / rasterizer_impl.cu computed for the
    same inputs (tests/golden/ref, generated by tests/golden/make_golden_ref.py through oracle/_ref).  Integer state
    (num_rendered, radii) is bit-exact; images 1e-5 (1e-4 is north_star's bound); gradients 1e-3 of the tensor max."""
    z, case, sc, cam, kw, dC, dF = _ref_case(path)
    c, f, r, g, st = util.run_oracle_b(sc, kw, dC, dF)
    assert st.num_rendered == int(z["num_rendered"])
    assert np.array_equal(r.numpy(), z["radii"])
    inc = case.get("include_feature", True)
    for got, ref in [(c, z["out_color"])] + ([(f, z["out_feat"])] if inc else []):
        rob, frag, frac = util.image_errors(got, torch.from_numpy(ref), st)
        assert rob <= 1e-5 and frag <= util.FRAGILE_TOL and frac <= util.FRAGILE_MAX_FRACTION
    if not inc:
        assert np.abs(z["out_feat"]).max() == 0 and np.abs(z["grad_language_feature"]).max() == 0
    fg = oracle_b.fragile_gaussians(st, 2e-5)
    for k in ("means2D", "means3D", "opacities", "colors_precomp", "language_feature", "cov3D", "sh", "scales",
              "rotations"):
        ref = torch.from_numpy(z[f"grad_{k}"])
        if ref.numel() == 0 or (k == "language_feature" and not inc):
            continue
        d = (g[k].reshape(ref.shape) - ref).abs().reshape(ref.shape[0], -1).max(1)[0]
        mx = ref.abs().max().item()
        assert (d[~fg].max().item() if (~fg).any() else 0.0) <= 1e-3 * mx + 1e-7, k
        assert (d[fg].max().item() if fg.any() else 0.0) <= util.FRAGILE_GRAD_TOL * mx + 1e-7, k


@pytest.mark.parametrize("path", [p for p in REF_GOLDEN if "128x128" not in p and "ragged" not in p],
                         ids=lambda p: os.path.basename(p)[:-4])
def test_oracle_a_matches_reference_kernels(path):
    """PIN: the independent autograd oracle against the reference's kernels (small cases; it is O(P * pixels))."""
    z, case, sc, cam, kw, dC, dF = _ref_case(path)
    ca, fa, ra, ga, aux = oracle_a.forward_backward(sc, types.SimpleNamespace(**kw), dC, dF)
    assert aux["num_rendered"] == int(z["num_rendered"]) and np.array_equal(ra.numpy(), z["radii"])
    assert np.abs(ca.numpy() - z["out_color"]).max() <= 1e-5
    if case.get("include_feature", True):
        assert np.abs(fa.numpy() - z["out_feat"]).max() <= 1e-5
    for k, v in ga.items():
        key = {"shs": "sh"}.get(k, k)
        if f"grad_{key}" not in z.files or v.numel() == 0:
            continue
        ref = z[f"grad_{key}"].reshape(v.shape)
        if key == "language_feature" and not case.get("include_feature", True):
            continue
        if key == "scales":
            # A third place where the reference's analytic backward is not the derivative of its forward: computeCov3D scales
            # S by `mod` (forward.cu:122-126), but the backward takes dL/dscale through dL/dM without it (backward.cu:295,
            # 325-327: "dL_dscale->x = dot(Rt[0], dL_dMt[0])").  Autograd's gradient is mod x the reference's; the product
            # (and Oracle B) reproduce the reference's.
            ref = ref * np.float32(case.get("scale_modifier", 1.0))
        assert np.abs(v.numpy() - ref).max() <= 1e-3 * np.abs(ref).max() + 1e-7, k
