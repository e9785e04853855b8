"""Run the reference's GeneralizableGSEmbedNet (agents/manigaussian_bc/models_embed.py, loaded by tests/ref_import.py and
executed unmodified) with probes on the tensors the HIP kernels replace -- TEST INFRASTRUCTURE.

One forward of that module contains, in order (models_embed.py:190-304):
  point_latent = grid_sample(dec_fts, canon(xyz))        | SURVEY 8f row 3 -> mgs_voxel_sample_pe_*   (manigaussian_amd.voxel)
  z_feature    = PositionalEncoding(canon(xyz))          |
  raw          = gs_parm_regresser(encoder(latent))      |   (GEMMs, torch)
  xyz/opacity/scale/rot/sh/feature maps = epilogue(raw)  | SURVEY 8f row 2 -> mgs_regress_epilogue_*   (manigaussian_amd.regressor)
  dyna_input   = cat(point_latent, maps.detach(), z, action)   | a14     -> mgs_deform_assemble_*      (manigaussian_amd.deform)
  delta        = gs_deformation_field(dyna_input)        | a15           -> ResnetFC (GEMMs) + mgs_mlp_*
  next.xyz/rot = xyz.detach() + d, normalize(rot.detach() + d) | a16     -> mgs_deform_apply_*
The probes capture point_latent, the encoder's input, raw, dyna_input and delta (each with retain_grad), so that one backward
through the reference yields the gradient every kernel's backward must reproduce.
"""
import types

import torch

import ref_import

OUT_KEYS = ("xyz_maps", "sh_maps", "rot_maps", "scale_maps", "opacity_maps", "feature_maps")
NEXT_KEYS = ("xyz_maps", "rot_maps")


def build_net(d_hidden=64, use_action=True, semantic=False, seed=0, device="cpu"):
    """GeneralizableGSEmbedNet at conf/method/ManiGaussian_BC.yaml's configuration (d_hidden a parameter), seeded; the
    parameters the reference initialises to zero (every bias, every fc_1 weight) are given values so that they matter."""
    ME = ref_import.load_models_embed()
    if ME is None:
        return None
    cfg = ref_import.method_cfg(d_hidden=d_hidden, use_dynamic_field=True, use_action=use_action,
                                foundation_model_name="diffusion" if semantic else None)
    torch.manual_seed(seed)
    net = ME.GeneralizableGSEmbedNet(cfg, with_gs_render=True)
    g = torch.Generator().manual_seed(seed + 1)
    with torch.no_grad():
        for n_, p in net.named_parameters():
            if "fc_1.weight" in n_:
                p.copy_(torch.randn(p.shape, generator=g) * (1.0 / p.shape[1]) ** 0.5)
            elif n_.endswith("bias"):
                p.copy_(torch.randn(p.shape, generator=g) * 0.1)
        # regressor head: raw outputs of order one, log-scales on both sides of the 0.05 clamp (ln 0.05 = -3.0)
        net.gs_parm_regresser.out.weight.mul_(0.5)
        net.gs_parm_regresser.out.bias[4:7] = -3.2
        net.gs_deformation_field.lin_out.weight.mul_(0.05)
    return net.to(device)


def make_inputs(N, D=8, seed=0, device="cpu", use_action=True):
    """data dict as NeuralRenderer.forward hands it to the embed net (neural_rendering.py:270-282)."""
    g = torch.Generator().manual_seed(seed)
    lo, hi = torch.tensor([-0.3, -0.5, 0.6]), torch.tensor([0.7, 0.5, 1.6])
    xyz = lo + (hi - lo) * (torch.rand(1, N, 3, generator=g) * 1.2 - 0.1)  # a tenth of the points outside the volume
    data = dict(xyz=xyz.to(device), dec_fts=torch.randn(1, 128, D, D + 1, D + 2, generator=g).to(device),
                lang=torch.randn(1, 128, generator=g).to(device), next={}, step=10000)
    if use_action:
        data["action"] = torch.randn(1, 8, generator=g).to(device)
    return data


def run(net, data, cotangents=None, seed=7):
    """forward (+ backward when cotangents is not False) of the reference module with probes.
    Returns a namespace: .data (the module's output dict), .probe (point_latent, latent_in, raw, dyna_input, delta),
    .cot (the cotangent per output), .grad (dec_fts, point_latent, raw, dyna_input, delta + every parameter)."""
    probe = {}
    data = dict(data)
    data["dec_fts"] = data["dec_fts"].detach().clone().requires_grad_(True)
    data["next"] = {}
    orig = net.sample_in_canonical_voxel

    def sample(xyz, voxel_feat):
        out = orig(xyz, voxel_feat)
        out.retain_grad()
        probe["point_latent"] = out
        return out

    def keep(name, t):
        if t.requires_grad:
            t.retain_grad()
        probe[name] = t

    net.sample_in_canonical_voxel = sample
    hooks = [
        net.encoder.register_forward_pre_hook(lambda m, a: keep("latent_in", a[0])),
        net.gs_parm_regresser.register_forward_hook(lambda m, a, o: keep("raw", o)),
        net.gs_deformation_field.register_forward_pre_hook(lambda m, a: keep("dyna_input", a[0])),
        net.gs_deformation_field.register_forward_hook(lambda m, a, o: keep("delta", o[0])),
    ]
    try:
        out = net(data)
    finally:
        for h in hooks:
            h.remove()
        del net.sample_in_canonical_voxel  # back to the class's method
    res = types.SimpleNamespace(data=out, probe=probe, cot={}, grad={})
    if cotangents is False:
        return res
    g = torch.Generator().manual_seed(seed)
    dev = out["xyz_maps"].device
    loss = 0.0
    for k in OUT_KEYS:
        res.cot[k] = (cotangents or {}).get(k, torch.randn(out[k].shape, generator=g).to(dev))
        loss = loss + (out[k] * res.cot[k]).sum()
    for k in NEXT_KEYS:
        res.cot["next_" + k] = (cotangents or {}).get("next_" + k, torch.randn(out["next"][k].shape, generator=g).to(dev))
        loss = loss + (out["next"][k] * res.cot["next_" + k]).sum()
    net.zero_grad(set_to_none=True)
    loss.backward()
    res.grad = {k: v.grad for k, v in probe.items() if v.grad is not None}
    res.grad["dec_fts"] = data["dec_fts"].grad
    res.grad.update({"param:" + n_: p.grad for n_, p in net.named_parameters() if p.grad is not None})
    return res
