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
import os

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


# ---- the same network with the elementwise passes fused (mgs_mlp.hip) and the GEMMs issued by hand ------------------

def tune_gemms(enable=True, max_ms=15, iterations=5, filename=None):
    """Let PyTorch's TunableOp pick the fp32 GEMM kernels (hipBLASLt or rocBLAS) for the MLP's shapes: the first call of
    every shape times the candidates (a few seconds in total), later calls use the winner.  Measured at configs[3]
    (M = 100 000, 512 x 512): forward 429 -> 373 us, data gradient 445 -> 363 us per GEMM, 17.9 -> 16.0 ms per step.  Same
    arithmetic type (fp32 on the matrix cores); process-wide PyTorch state, hence opt-in."""
    import torch.cuda.tunable as tunable
    tunable.enable(bool(enable))
    if enable:
        tunable.tuning_enable(True)
        tunable.set_max_tuning_duration(int(max_ms))
        tunable.set_max_tuning_iterations(int(iterations))
        if filename is None and "PYTORCH_TUNABLEOP_FILENAME" not in os.environ:
            import tempfile  # TunableOp writes its table at exit: not into the working directory unless asked to
            filename = os.path.join(tempfile.gettempdir(), f"mgs_tunableop_{os.getpid()}.csv")
        if filename is not None:
            tunable.set_filename(filename)


def _mlp_ok(n):
    return n % 4 == 0 and (n // 4) <= 256 and 256 % (n // 4) == 0


def _relu_bias(x, bias, want_relu=True, want_xb=True):
    """(relu(x), x + bias) in one pass over x."""
    M, N = x.shape
    a = torch.empty_like(x) if want_relu else None
    xb = torch.empty_like(x) if want_xb else None
    _lib.check(_lib.lib().mgs_mlp_relu_bias(M, N, x.data_ptr(), 0 if bias is None else bias.data_ptr(),
                                            0 if a is None else a.data_ptr(), 0 if xb is None else xb.data_ptr(),
                                            _stream(x.device)), "mgs_mlp_relu_bias")
    return a, xb


def _relu_backward(g_pre, act, g_res, colsum):
    """g_pre * (act > 0) [+ g_res] written over g_pre; colsum (zeroed [N]) accumulates the column sums of the result."""
    M, N = g_pre.shape
    _lib.check(_lib.lib().mgs_mlp_relu_backward(M, N, g_pre.data_ptr(), act.data_ptr(),
                                                0 if g_res is None else g_res.data_ptr(), g_pre.data_ptr(),
                                                0 if colsum is None else colsum.data_ptr(), _stream(g_pre.device)),
               "mgs_mlp_relu_backward")
    return g_pre


def _addmm_relu(bias, a, wt):
    """relu(a @ wt + bias) with bias and ReLU in the GEMM epilogue where this torch has it."""
    f = getattr(torch, "_addmm_activation", None)
    return f(bias, a, wt) if f is not None else torch.addmm(bias, a, wt).relu_()


_WGRAD_SPLIT = 8        # batches of the split-K weight gradient
_WGRAD_MIN_ROWS = 4096  # ... taken when each batch still has this many rows


def _wgrad(g, x):
    """g^T @ x ([M, No], [M, K] -> [No, K]).  The output is small and the reduction long (M rows): split the rows into
    batches so that the GEMM fills the chip, then add the partial products."""
    M = g.shape[0]
    S = _WGRAD_SPLIT
    if S > 1 and M % S == 0 and M >= _WGRAD_MIN_ROWS * S:
        return torch.bmm(g.view(S, M // S, -1).transpose(1, 2), x.view(S, M // S, -1)).sum(0)
    return g.t() @ x


class _FusedResnetFC(torch.autograd.Function):
    """ResnetFC.forward / backward with hand-issued GEMMs (torch -> hipBLASLt) and fused elementwise passes:
    * the residual stream picks up the biases that are added to it next (fc_1's, lin_z's) in the pass that computes its
      ReLU, so that fc_1 and lin_z accumulate into it inside the GEMM (beta = 1) -- no separate add pass;
    * fc_0's bias and ReLU ride in the GEMM epilogue (torch._addmm_activation);
    * backward: ReLU mask, residual add and the bias gradient (column sums) in one pass; weight gradients as split-K
      batched GEMMs.
    Same arithmetic as the module's plain path up to the association of the bias additions."""

    @staticmethod
    def forward(ctx, zx, d_latent, n_blocks, n_lin_z, *params):
        W_in, b_in, W_out, b_out = params[:4]
        blk = [params[4 + 4 * i: 8 + 4 * i] for i in range(n_blocks)]                    # (W0, b0, W1, b1)
        lz = [params[4 + 4 * n_blocks + 2 * i: 6 + 4 * n_blocks + 2 * i] for i in range(n_lin_z)]  # (Wz, bz)
        z = zx[:, :d_latent].contiguous()
        xin = zx[:, d_latent:].contiguous()
        s = torch.addmm(b_in + lz[0][1] if n_lin_z > 0 else b_in, xin, W_in.t())
        if n_lin_z > 0:
            s.addmm_(z, lz[0][0].t())
        saved = []
        for i in range(n_blocks):
            W0, b0, W1, b1 = blk[i]
            a, xb = _relu_bias(s, b1 + lz[i + 1][1] if i + 1 < n_lin_z else b1)
            h = _addmm_relu(b0, a, W0.t())
            xb.addmm_(h, W1.t())
            if i + 1 < n_lin_z:
                xb.addmm_(z, lz[i + 1][0].t())
            saved += [s, a, h]
            s = xb
        a_out, _ = _relu_bias(s, None, want_xb=False)
        delta = torch.addmm(b_out, a_out, W_out.t())
        ctx.save_for_backward(z, xin, s, a_out, *saved, *params)
        ctx.cfg = (d_latent, n_blocks, n_lin_z)
        ctx.set_materialize_grads(False)
        return delta, s

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g_delta, g_x):
        d_latent, n_blocks, n_lin_z = ctx.cfg
        t = ctx.saved_tensors
        z, xin, x_last, a_out = t[:4]
        saved = t[4: 4 + 3 * n_blocks]
        params = t[4 + 3 * n_blocks:]
        W_in, b_in, W_out, b_out = params[:4]
        blk = [params[4 + 4 * i: 8 + 4 * i] for i in range(n_blocks)]
        lz = [params[4 + 4 * n_blocks + 2 * i: 6 + 4 * n_blocks + 2 * i] for i in range(n_lin_z)]
        H = W_in.shape[0]
        dev = z.device
        grads = [None] * len(params)
        need = ctx.needs_input_grad[4:]  # per parameter: a frozen weight's gradient GEMM is skipped

        def wgrad(k, g, x):
            if need[k]:
                grads[k] = _wgrad(g, x)

        cs = torch.zeros(H, device=dev)
        if g_delta is not None:
            g_delta = g_delta.contiguous()
            wgrad(2, g_delta, a_out)
            if need[3]:
                grads[3] = g_delta.sum(0)
            g = _relu_backward(g_delta @ W_out, x_last, None if g_x is None else g_x.contiguous(), cs)  # dL/d(last block's out)
        else:  # only the features were used
            g = g_x.contiguous().clone()
            cs = g.sum(0)
        need_in = ctx.needs_input_grad[0]
        gz = torch.zeros_like(z) if (need_in and n_lin_z > 0) else None
        for i in reversed(range(n_blocks)):
            W0, b0, W1, b1 = blk[i]
            s_i, a_i, h_i = saved[3 * i: 3 * i + 3]
            k = 4 + 4 * i
            grads[k + 3] = cs                                 # b1: column sums of dL/d out_i
            if i + 1 < n_lin_z:                               # lin_z[i+1] fed the same sum
                kz = 4 + 4 * n_blocks + 2 * (i + 1)
                wgrad(kz, g, z)
                grads[kz + 1] = cs.clone()
                if gz is not None:
                    gz.addmm_(g, lz[i + 1][0])
            wgrad(k + 2, g, h_i)
            db0 = torch.zeros(H, device=dev)
            gh = _relu_backward(g @ W1, h_i, None, db0)
            grads[k + 1] = db0
            wgrad(k, gh, a_i)
            cs = torch.zeros(H, device=dev)
            g = _relu_backward(gh @ W0, s_i, g, cs)            # dL/d s_i = dL/d(out of block i-1, + lin_z[i])
        if n_lin_z > 0:
            kz = 4 + 4 * n_blocks
            wgrad(kz, g, z)
            grads[kz + 1] = cs.clone()
            if gz is not None:
                gz.addmm_(g, lz[0][0])
        wgrad(0, g, xin)
        grads[1] = cs
        grads = [g_ if n_ else None for g_, n_ in zip(grads, need)]
        g_zx = None
        if need_in:
            g_xin = g @ W_in
            g_zx = torch.cat([gz if gz is not None else torch.zeros_like(z), g_xin], 1)
        return (g_zx, None, None, None, *grads)


class ResnetFC(nn.Module):
    """resnetfc.py:65-177 with combine_layer >= n_blocks, use_spade False, beta 0 (the deformation-field
    configuration).  zx = [z (d_latent) | x (d_in)].  fp32 tensors on a GPU take the fused path (_FusedResnetFC); anything
    else (CPU, autocast) the plain torch ops below -- the same network either way."""

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
        self.fused = True

    def _fusable(self, zx):
        return (self.fused and zx.is_cuda and zx.dtype == torch.float32 and zx.dim() == 2 and _mlp_ok(self.d_hidden)
                and not torch.is_autocast_enabled() and self.lin_in.weight.dtype == torch.float32)

    def forward(self, zx):
        assert zx.size(-1) == self.d_latent + self.d_in, f"{zx.size(-1)} != {self.d_latent} + {self.d_in}"
        if self._fusable(zx):
            n_lin_z = min(self.combine_layer, len(self.lin_z), self.n_blocks)
            params = [self.lin_in.weight, self.lin_in.bias, self.lin_out.weight, self.lin_out.bias]
            for b in self.blocks:
                params += [b.fc_0.weight, b.fc_0.bias, b.fc_1.weight, b.fc_1.bias]
            for i in range(n_lin_z):
                params += [self.lin_z[i].weight, self.lin_z[i].bias]
            return _FusedResnetFC.apply(zx, self.d_latent, self.n_blocks, n_lin_z, *params)
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
