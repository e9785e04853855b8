"""Golden vectors of the deformation field from the REFERENCE's own ResnetFC (agents/manigaussian_bc/resnetfc.py:65-177),
run here on CPU from /root/reference (tests/ref_import.py) at the configuration of conf/method/ManiGaussian_BC.yaml:146-157
with a narrower hidden layer (64 instead of 512) so that the fixture stays small; the live test in
tests/test_deform_mlp.py covers d_hidden = 512 whenever /root/reference is present.

  python tests/golden/make_golden_deform.py      ->  tests/golden/deform/deform_h64_{action,semantic}.npz

Pipeline restated around the module exactly as models_embed.py:255-304 calls it (input assembly by torch.cat,
combine_inner_dims=(1, N), split [3, 4], xyz + delta, normalize(rot + delta))."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import ref_import  # noqa: E402


def reference_pipeline(mlp, lat, z, xyz, sh, rot, scale, op, feat, action):
    """models_embed.py:255-304 with the reference module `mlp`; returns (next_xyz [1,N,3], next_rot [1,N,4])."""
    N = lat.shape[0]
    parts = [lat, xyz.detach().reshape(N, 3), sh[:, 0].detach().reshape(N, 3), sh[:, 1:].detach().reshape(N, 9),
             rot.detach().reshape(N, 4), scale.detach().reshape(N, 3), op.detach().reshape(N, 1)]
    if feat is not None:
        parts.append(feat.detach().reshape(N, 3))
    parts.append(z)
    dyna = torch.cat(parts, dim=-1)
    if action is not None:
        dyna = torch.cat((dyna, action.repeat(N, 1)), dim=-1)
    out, _ = mlp(dyna, combine_inner_dims=(1, N), combine_index=None, dim_size=None, language_embed=None, batch_size=1)
    dxyz, drot = out.split([3, 4], dim=-1)
    return xyz.detach().reshape(1, N, 3) + dxyz, torch.nn.functional.normalize(rot.detach().reshape(1, N, 4) + drot, dim=-1)


def make(name, use_action, use_semantic, seed, d_hidden=64, N=192):
    R = ref_import.load_resnetfc()
    assert R is not None, "needs /root/reference"
    g = torch.Generator().manual_seed(seed)
    rn = lambda *s: torch.randn(*s, generator=g)
    d_in = 23 + 39 + (8 if use_action else 0) + (3 if use_semantic else 0)
    torch.manual_seed(seed)
    mlp = R.ResnetFC(d_in=d_in, d_latent=128, d_lang=128, d_out=7, d_hidden=d_hidden, n_blocks=5, combine_layer=3,
                     beta=0.0, use_spade=False)
    with torch.no_grad():  # the reference zero-initialises fc_1 and every bias: give them values so they are exercised
        for n_, p in mlp.named_parameters():
            if "fc_1.weight" in n_:
                p.copy_(rn(*p.shape) * (1.0 / d_hidden) ** 0.5)
            elif n_.endswith("bias"):
                p.copy_(rn(*p.shape) * 0.1)
    lat, z = rn(N, 128).requires_grad_(True), rn(N, 39).requires_grad_(True)
    xyz, sh, rot, scale, op = rn(N, 3), rn(N, 4, 3), rn(N, 4), rn(N, 3).abs() * 0.02, torch.sigmoid(rn(N, 1))
    feat = rn(N, 3) if use_semantic else None
    action = rn(1, 8) if use_action else None
    nx, nr = reference_pipeline(mlp, lat, z, xyz, sh, rot, scale, op, feat, action)
    wx, wr = rn(1, N, 3), rn(1, N, 4)
    params = list(mlp.parameters())
    grads = torch.autograd.grad((nx * wx).sum() + (nr * wr).sum(), [lat, z] + params)
    out = dict(case=np.frombuffer(repr(dict(use_action=use_action, use_semantic=use_semantic, d_hidden=d_hidden, N=N,
                                            d_in=d_in)).encode(), dtype=np.uint8))
    for k, v in dict(lat=lat, z=z, xyz=xyz, sh=sh, rot=rot, scale=scale, op=op, wx=wx, wr=wr).items():
        out["in_" + k] = v.detach().numpy()
    if feat is not None:
        out["in_feat"] = feat.numpy()
    if action is not None:
        out["in_action"] = action.numpy()
    for n_, p in mlp.state_dict().items():
        out["sd_" + n_] = p.numpy()
    out["out_xyz"], out["out_rot"] = nx.detach().numpy(), nr.detach().numpy()
    out["grad_lat"], out["grad_z"] = grads[0].numpy(), grads[1].numpy()
    for (n_, _), g_ in zip(mlp.named_parameters(), grads[2:]):
        out["gradp_" + n_] = g_.numpy()
    os.makedirs(os.path.join(HERE, "deform"), exist_ok=True)
    np.savez_compressed(os.path.join(HERE, "deform", name + ".npz"), **out)
    print(name, "params", sum(p.numel() for p in params), "d_in", d_in)


if __name__ == "__main__":
    make("deform_h64_action", True, False, 1)
    make("deform_h64_semantic", True, True, 2)
