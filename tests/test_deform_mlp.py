"""the deformation MLP (manigaussian_amd.deform.ResnetFC / DeformationField) against the REFERENCE's
own module agents/manigaussian_bc/resnetfc.py:65-177 -- shared weights, outputs and gradients.

  * live (development container, /root/reference present): the reference class is imported in place
    (tests/ref_import.py) at the production size (d_hidden 512, conf/method/ManiGaussian_BC.yaml:146-157);
  * fixtures (everywhere, incl. the GPU box): tests/golden/deform/*.npz, generated from the same reference class by
    tests/golden/make_golden_deform.py (d_hidden 64 to keep the files small).
"""
import glob
import os

import numpy as np
import pytest
import torch

import ref_import
from manigaussian_amd import deform

GOLDEN = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "deform", "*.npz")))


def _load(path):
    z = np.load(path)
    case = eval(bytes(z["case"]).decode())
    t = lambda k: torch.from_numpy(z[k])
    ins = {k[3:]: t(k) for k in z.files if k.startswith("in_")}
    sd = {k[3:]: t(k) for k in z.files if k.startswith("sd_")}
    gp = {k[6:]: t(k) for k in z.files if k.startswith("gradp_")}
    return case, ins, sd, gp, z


def _torch_pipeline(mlp, lat, z, xyz, sh, rot, scale, op, feat, action):
    """The torch ops of models_embed.py:255-304 around OUR ResnetFC (CPU stand-in for the HIP assembly/apply kernels,
    which test_gpu_parity.py::test_deform_apply_and_assembly_match_torch pins bit for bit against these ops)."""
    N = lat.shape[0]
    parts = [lat, xyz, sh[:, 0], sh[:, 1:].reshape(N, 9), rot, scale, op] + ([feat] if feat is not None else []) + [z]
    if action is not None:
        parts.append(action.repeat(N, 1))
    delta, _ = mlp(torch.cat(parts, -1))
    return xyz + delta[:, :3], torch.nn.functional.normalize(rot + delta[:, 3:], dim=-1)


def _check(got, ref, tol, what):
    scale = max(float(ref.abs().max()), 1e-12)
    err = float((got.detach().reshape(ref.shape) - ref).abs().max())
    assert err <= tol * scale + 1e-9, f"{what}: {err:.3e} vs max {scale:.3e}"


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p)[:-4] for p in GOLDEN])
def test_resnetfc_matches_reference_fixture(path):
    """State-dict compatible (strict load) and numerically the reference module: outputs 1e-6, gradients 1e-5 of max."""
    case, ins, sd, gp, z = _load(path)
    mlp = deform.ResnetFC(case["d_in"], d_out=7, n_blocks=5, d_latent=128, d_hidden=case["d_hidden"], combine_layer=3)
    mlp.load_state_dict(sd, strict=True)
    lat, zf = ins["lat"].clone().requires_grad_(True), ins["z"].clone().requires_grad_(True)
    nx, nr = _torch_pipeline(mlp, lat, zf, ins["xyz"], ins["sh"], ins["rot"], ins["scale"], ins["op"], ins.get("feat"),
                             ins.get("action"))
    _check(nx, torch.from_numpy(z["out_xyz"]), 1e-6, "next xyz")
    _check(nr, torch.from_numpy(z["out_rot"]), 1e-6, "next rot")
    params = dict(mlp.named_parameters())
    grads = torch.autograd.grad((nx * ins["wx"][0]).sum() + (nr * ins["wr"][0]).sum(), [lat, zf] + list(params.values()))
    _check(grads[0], torch.from_numpy(z["grad_lat"]), 1e-5, "grad point_latent")
    _check(grads[1], torch.from_numpy(z["grad_z"]), 1e-5, "grad z_feature")
    assert set(params) == set(gp)
    for (n_, _), g_ in zip(params.items(), grads[2:]):
        _check(g_, gp[n_], 1e-5, f"grad {n_}")


@pytest.mark.skipif(not ref_import.have_reference(), reason="/root/reference not present (GPU box): fixtures cover it")
@pytest.mark.parametrize("use_semantic", [False, True], ids=["d_in_70", "d_in_73_semantic"])
def test_resnetfc_matches_reference_module_live(use_semantic):
    """The reference class itself at the production size: load ITS state dict into ours, same inputs, outputs and every
    gradient within 1e-6 / 1e-5 of the tensor's max (both run the same torch CPU GEMMs)."""
    R = ref_import.load_resnetfc()
    torch.manual_seed(11)
    d_in = 23 + 39 + 8 + (3 if use_semantic else 0)
    ref = R.ResnetFC(d_in=d_in, d_latent=128, d_lang=128, d_out=7, d_hidden=512, n_blocks=5, combine_layer=3, beta=0.0,
                     use_spade=False)
    with torch.no_grad():
        for n_, p in ref.named_parameters():
            if "fc_1.weight" in n_:
                p.normal_(0, (1.0 / 512) ** 0.5)
            elif n_.endswith("bias"):
                p.normal_(0, 0.1)
    ours = deform.ResnetFC(d_in, d_out=7, n_blocks=5, d_latent=128, d_hidden=512, combine_layer=3)
    ours.load_state_dict(ref.state_dict(), strict=True)
    assert [k for k, _ in ours.named_parameters()] == [k for k, _ in ref.named_parameters()]
    N = 96
    zx = torch.randn(N, 128 + d_in)
    a, b = zx.clone().requires_grad_(True), zx.clone().requires_grad_(True)
    o_ref, h_ref = ref(a, combine_inner_dims=(1, N))      # the call shape of models_embed.py:289-296
    o, h = ours(b)
    _check(o, o_ref, 1e-6, "delta")
    _check(h, h_ref, 1e-6, "last hidden")
    w = torch.randn(N, 7)
    g_ref = torch.autograd.grad((o_ref.reshape(N, 7) * w).sum(), [a] + list(ref.parameters()))
    g = torch.autograd.grad((o * w).sum(), [b] + list(ours.parameters()))
    for x, y, (n_, _) in zip(g, g_ref, [("zx", None)] + list(ref.named_parameters())):
        _check(x, y, 1e-5, f"grad {n_}")


def test_fresh_initialisation_follows_the_reference():
    """resnetfc.py:33-37,94-121: zero biases, zero fc_1 weights, kaiming fan-in elsewhere (std = sqrt(2 / fan_in))."""
    torch.manual_seed(0)
    m = deform.ResnetFC(70, d_hidden=512)
    for n_, p in m.named_parameters():
        if n_.endswith("bias") or "fc_1.weight" in n_:
            assert float(p.abs().max()) == 0.0, n_
        else:
            std = (2.0 / p.shape[1]) ** 0.5
            assert abs(float(p.std()) / std - 1.0) < 0.1, n_


@pytest.mark.gpu
@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p)[:-4] for p in GOLDEN])
def test_deformation_field_on_gpu_matches_reference_fixture(path):
    """DeformationField end to end on the MI355X (HIP input assembly -> MLP GEMMs -> HIP apply epilogue) against the
    reference module's outputs and gradients.  fp32 GEMMs on the GPU sum in another order than the CPU's: 2e-5 / 1e-4."""
    assert torch.cuda.is_available()
    dev = torch.device("cuda:0")
    case, ins, sd, gp, z = _load(path)
    field = deform.DeformationField(d_latent=128, d_z=39, use_action=case["use_action"],
                                    use_semantic_feature=case["use_semantic"], d_hidden=case["d_hidden"]).to(dev)
    field.mlp.load_state_dict(sd, strict=True)
    d = {k: v.to(dev) for k, v in ins.items()}
    lat, zf = d["lat"].clone().requires_grad_(True), d["z"].clone().requires_grad_(True)
    nxt = field(lat, zf, d["xyz"], d["sh"], d["rot"], d["scale"], d["op"], feature=d.get("feat"), action=d.get("action"))
    _check(nxt["xyz"].cpu(), torch.from_numpy(z["out_xyz"]), 2e-5, "next xyz")
    _check(nxt["rot"].cpu(), torch.from_numpy(z["out_rot"]), 2e-5, "next rot")
    params = dict(field.mlp.named_parameters())
    grads = torch.autograd.grad((nxt["xyz"] * d["wx"][0]).sum() + (nxt["rot"] * d["wr"][0]).sum(),
                                [lat, zf] + list(params.values()))
    _check(grads[0].cpu(), torch.from_numpy(z["grad_lat"]), 1e-4, "grad point_latent")
    _check(grads[1].cpu(), torch.from_numpy(z["grad_z"]), 1e-4, "grad z_feature")
    for (n_, _), g_ in zip(params.items(), grads[2:]):
        _check(g_.cpu(), gp[n_], 1e-4, f"grad {n_}")


@pytest.mark.gpu
@pytest.mark.parametrize("M,hidden,use_x", [(1000, 512, True), (8192, 512, False), (4100, 64, True)])
def test_fused_resnetfc_equals_the_plain_module_on_gpu(M, hidden, use_x):
    """The fused path (hand-issued GEMMs + mgs_mlp.hip elementwise passes, split-K weight gradients in the 8192-row case) and the
    module's plain torch path, both in fp32 with the same non-trivial weights, against the plain path in float64: outputs,
    input gradient and every parameter gradient -- the fused path must be as close to the exact result as the plain one."""
    dev = torch.device("cuda:0")
    torch.manual_seed(M)
    deform._WGRAD_MIN_ROWS = 512 if M == 8192 else 4096
    m = deform.ResnetFC(70, d_hidden=hidden).to(dev)
    with torch.no_grad():
        for p in m.parameters():  # the reference's initialisation zeroes fc_1 and the biases: exercise them
            p.copy_(torch.randn_like(p) * (0.5 / p.shape[-1] ** 0.5 if p.dim() == 2 else 0.1))
    zx = torch.randn(M, 198, device=dev)
    wd, wx = torch.randn(M, 7, device=dev), torch.randn(M, hidden, device=dev) / hidden

    def run(mod, fused, dt):
        mod.fused = fused
        zin = zx.to(dt).requires_grad_(True)
        delta, x = mod(zin)
        loss = (delta * wd.to(dt)).sum() + ((x * wx.to(dt)).sum() if use_x else 0.0)
        grads = torch.autograd.grad(loss, [zin] + list(mod.parameters()))
        return [delta.detach().double(), x.detach().double()] + [g.double() for g in grads]

    import copy
    exact = run(copy.deepcopy(m).double(), False, torch.float64)
    plain, fused = run(m, False, torch.float32), run(m, True, torch.float32)
    names = ["delta", "x", "grad zx"] + [f"grad {n}" for n, _ in m.named_parameters()]
    deform._WGRAD_MIN_ROWS = 4096
    # A pre-activation within an ulp of zero can land on either side of the ReLU in either fp32 path (the fused one adds
    # the biases in another order): that row's gradients then differ by a whole term (expected: one flip per ~1e7
    # activations).  So: 99 % of the elements of every tensor are as close to float64 as the plain path's (x5) or 1e-4 of
    # the tensor's scale, and every element is within 5 %.
    for n_, f, p, e in zip(names, fused, plain, exact):
        scale = float(e.abs().max()) + 1e-12
        ef, ep = (f - e).abs().flatten(), (p - e).abs().flatten()
        q = (lambda v: float(torch.quantile(v[:: max(1, v.numel() // 1000000)], 0.99))) if ef.numel() > 100 else (lambda v: float(v.max()))
        assert q(ef) <= max(5.0 * q(ep), 1e-4 * scale) + 1e-6, (n_, q(ef), q(ep), scale)
        assert float(ef.max()) <= 5e-2 * scale + 1e-6, (n_, float(ef.max()), scale)
