"""Deformation-field per-Gaussian apply (reference: agents/manigaussian_bc/models_embed.py:255-304).

Three pieces:
  assemble_deform_input(...)  fused input assembly for gs_deformation_field (models_embed.py:258-287):
                              one HIP pass instead of 9 detached views + two torch.cat + a repeat.
  deform_apply(...)           next.xyz = xyz.detach() + dxyz, next.rot = normalize(rot.detach() + drot)
                              (models_embed.py:295-299), fused fwd+bwd HIP kernels.
  ResnetFC / DeformationField mirror of agents/manigaussian_bc/resnetfc.py:65-177 as configured by
                              conf/method/ManiGaussian_BC.yaml:146-157 (d_in 73|70, d_latent 128, 512x5,
                              d_out 7, ReLU).  The GEMMs stay in torch (hipBLASLt / MFMA): they are dense
                              contractions, not part of the hand-written path.
"""
import ctypes

import torch
import torch.nn as nn

from . import _lib


def _stream(dev):
    return ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)


def _c(t):
    """contiguous float32 (the kernels are fp32; a bf16/fp16 tensor from an autocast MLP is widened, its gradient is
    narrowed back by autograd)."""
    if not t.is_floating_point():
        raise RuntimeError(f"expected a floating-point tensor, got {t.dtype}")
    return t.float().contiguous()


class _AssembleDeformInput(torch.autograd.Function):
    @staticmethod
    def forward(ctx, point_latent, z_feature, xyz, sh, rot, scale, opacity, feature, action):
        L = _lib.lib()
        dev = point_latent.device
        N, DL = point_latent.shape
        DZ = z_feature.shape[1]
        DA = 0 if action is None else action.shape[-1]
        has_feat = feature is not None
        stride = DL + 23 + (3 if has_feat else 0) + DZ + DA
        out = torch.empty((N, stride), dtype=torch.float32, device=dev)
        t = [_c(point_latent), _c(xyz.reshape(N, 3)), _c(sh.reshape(N, 12)), _c(rot.reshape(N, 4)),
             _c(scale.reshape(N, 3)), _c(opacity.reshape(N, 1))]
        f = _c(feature.reshape(N, 3)) if has_feat else None
        z = _c(z_feature)
        a = _c(action.reshape(-1)) if DA else None
        with torch.cuda.device(dev):
            _lib.check(L.mgs_deform_assemble_forward(
                N, DL, DZ, DA, t[0].data_ptr(), t[1].data_ptr(), t[2].data_ptr(), t[3].data_ptr(), t[4].data_ptr(),
                t[5].data_ptr(), f.data_ptr() if has_feat else None, z.data_ptr(), a.data_ptr() if DA else None,
                out.data_ptr(), _stream(dev)), "deform_assemble_forward")
        ctx.dims = (N, DL, DZ, DA, has_feat)
        return out

    @staticmethod
    def backward(ctx, g_out):
        L = _lib.lib()
        N, DL, DZ, DA, has_feat = ctx.dims
        dev = g_out.device
        g_out = _c(g_out)
        g_lat = torch.empty((N, DL), dtype=torch.float32, device=dev)
        g_z = torch.empty((N, DZ), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            _lib.check(L.mgs_deform_assemble_backward(N, DL, DZ, DA, int(has_feat), g_out.data_ptr(), g_lat.data_ptr(),
                                                      g_z.data_ptr(), _stream(dev)), "deform_assemble_backward")
        # everything but point_latent and z_feature is .detach()ed in the reference
        return g_lat, g_z, None, None, None, None, None, None, None


def assemble_deform_input(point_latent, z_feature, xyz, sh, rot, scale, opacity, feature=None, action=None):
    """dyna_input [N, DL + 23 (+3) + DZ (+DA)] (models_embed.py:258-287).  Gradient reaches point_latent and
    z_feature only.  sh is [N,4,3] (f_dc, f_rest); feature only when foundation_model_name == 'diffusion'."""
    return _AssembleDeformInput.apply(point_latent, z_feature, xyz, sh, rot, scale, opacity, feature, action)


class _DeformApply(torch.autograd.Function):
    @staticmethod
    def forward(ctx, delta, xyz, rot):
        L = _lib.lib()
        dev = delta.device
        N = delta.shape[0]
        delta, xyz, rot = _c(delta), _c(xyz.reshape(N, 3)), _c(rot.reshape(N, 4))
        xyz_out = torch.empty((N, 3), dtype=torch.float32, device=dev)
        rot_out = torch.empty((N, 4), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            _lib.check(L.mgs_deform_apply_forward(N, xyz.data_ptr(), rot.data_ptr(), delta.data_ptr(),
                                                  xyz_out.data_ptr(), rot_out.data_ptr(), _stream(dev)),
                       "deform_apply_forward")
        ctx.save_for_backward(delta, rot)
        return xyz_out, rot_out

    @staticmethod
    def backward(ctx, g_xyz, g_rot):
        L = _lib.lib()
        delta, rot = ctx.saved_tensors
        dev = delta.device
        N = delta.shape[0]
        g_xyz = _c(g_xyz) if g_xyz is not None else torch.zeros((N, 3), dtype=torch.float32, device=dev)
        g_rot = _c(g_rot) if g_rot is not None else torch.zeros((N, 4), dtype=torch.float32, device=dev)
        g_delta = torch.empty((N, 7), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            _lib.check(L.mgs_deform_apply_backward(N, rot.data_ptr(), delta.data_ptr(), g_xyz.data_ptr(),
                                                   g_rot.data_ptr(), g_delta.data_ptr(), _stream(dev)),
                       "deform_apply_backward")
        return g_delta, None, None


def deform_apply(delta, xyz, rot):
    """(next_xyz [N,3], next_rot [N,4]) from the MLP output delta [N,7]; xyz/rot are treated as detached."""
    return _DeformApply.apply(delta, xyz, rot)


# ---- the MLP itself: torch GEMMs, architecture restated from resnetfc.py -------------------------

class ResnetBlockFC(nn.Module):
    """x + fc_1(relu(fc_0(relu(x))))  (resnetfc.py:10-62, beta = 0 -> ReLU, size_in == size_out)."""

    def __init__(self, size):
        super().__init__()
        self.fc_0 = nn.Linear(size, size)
        self.fc_1 = nn.Linear(size, size)
        nn.init.constant_(self.fc_0.bias, 0.0)
        nn.init.kaiming_normal_(self.fc_0.weight, a=0, mode="fan_in")
        nn.init.constant_(self.fc_1.bias, 0.0)
        nn.init.zeros_(self.fc_1.weight)

    def forward(self, x):
        return x + self.fc_1(torch.relu(self.fc_0(torch.relu(x))))


class ResnetFC(nn.Module):
    """resnetfc.py:65-177 with combine_layer >= n_blocks, use_spade False, beta 0 (the deformation-field
    configuration).  zx = [z (d_latent) | x (d_in)]."""

    def __init__(self, d_in, d_out=7, n_blocks=5, d_latent=128, d_hidden=512, combine_layer=3):
        super().__init__()
        self.d_in, self.d_out, self.d_latent, self.d_hidden, self.n_blocks = d_in, d_out, d_latent, d_hidden, n_blocks
        self.combine_layer = combine_layer
        self.lin_in = nn.Linear(d_in, d_hidden)
        self.lin_out = nn.Linear(d_hidden, d_out)
        self.blocks = nn.ModuleList([ResnetBlockFC(d_hidden) for _ in range(n_blocks)])
        n_lin_z = min(combine_layer, n_blocks)
        self.lin_z = nn.ModuleList([nn.Linear(d_latent, d_hidden) for _ in range(n_lin_z)])
        for lin in [self.lin_in, self.lin_out, *self.lin_z]:
            nn.init.constant_(lin.bias, 0.0)
            nn.init.kaiming_normal_(lin.weight, a=0, mode="fan_in")

    def forward(self, zx):
        assert zx.size(-1) == self.d_latent + self.d_in, f"{zx.size(-1)} != {self.d_latent} + {self.d_in}"
        z, x = zx[..., : self.d_latent], zx[..., self.d_latent:]
        x = self.lin_in(x)
        for i in range(self.n_blocks):
            # combine_interleaved over a size-1 view dimension is the identity (utils.py:121-131)
            if i < self.combine_layer and i < len(self.lin_z):
                x = x + self.lin_z[i](z)
            x = self.blocks[i](x)
        return self.lin_out(torch.relu(x)), x


class DeformationField(nn.Module):
    """gs_deformation_field + input assembly + apply (models_embed.py:98-112, 255-304)."""

    def __init__(self, d_latent=128, d_z=39, use_action=True, use_semantic_feature=False, d_hidden=512, n_blocks=5,
                 combine_layer=3):
        super().__init__()
        self.use_action, self.use_semantic_feature = use_action, use_semantic_feature
        d_in = 23 + d_z + (8 if use_action else 0) + (3 if use_semantic_feature else 0)  # 70 / 73
        self.mlp = ResnetFC(d_in, d_out=7, n_blocks=n_blocks, d_latent=d_latent, d_hidden=d_hidden,
                            combine_layer=combine_layer)

    def forward(self, point_latent, z_feature, xyz, sh, rot, scale, opacity, feature=None, action=None):
        zx = assemble_deform_input(point_latent, z_feature, xyz, sh, rot, scale, opacity,
                                   feature if self.use_semantic_feature else None,
                                   action if self.use_action else None)
        delta, _ = self.mlp(zx)
        next_xyz, next_rot = deform_apply(delta, xyz, rot)
        # the rest passes through detached (models_embed.py:300-304)
        return dict(xyz=next_xyz, rot=next_rot, sh=sh.detach(), scale=scale.detach(), opacity=opacity.detach(),
                    feature=None if feature is None else feature.detach())
