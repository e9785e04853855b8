"""Golden vectors from the REFERENCE's own Python modules either side of the rasterizer, executed unmodified on CPU from
/root/reference through tests/ref_import.py (stand-ins only for termcolor / visdom / dotmap / torchvision / `.attention`):

  tests/golden/mg/embed_h32_{action,semantic}.npz
      agents/manigaussian_bc/models_embed.py:GeneralizableGSEmbedNet.forward (lines 190-304) at the configuration of
      conf/method/ManiGaussian_BC.yaml:113-162 with d_hidden = 32 (fixture size; the live tests run 512), N = 129 points,
      a 6 x 7 x 8 voxel volume: inputs, the module's state dict, every output map, and the probed intermediates
      (point_latent, the encoder's input [point_latent | positional code], the regressor's raw 26-vector, dyna_input, the
      deformation MLP's delta) with the gradients a backward through the reference leaves on them.
      Pins SURVEY.md 8f row 2 (mgs_regress_epilogue_*), 8f row 3 (mgs_voxel_sample_*), 8a a14 (mgs_deform_assemble_*),
      a15 (ResnetFC) and a16 (mgs_deform_apply_*).
  tests/golden/mg/novel_calib.npz
      agents/manigaussian_bc/neural_rendering.py:NeuralRenderer.get_novel_calib (lines 205-248) with
      graphics_utils.py:17-53 under it, for look-at cameras of both focal signs, off-centre principal points, two image
      sizes and a non-trivial (trans, scale).  Pins SURVEY.md 8f row 4 (mgs_novel_calib, mgs_novel_calib_host).
      NumPy note: the reference was written for NumPy 1.x, where `python_float / np.float32` is evaluated in float64; under
      NumPy 2 (NEP 50) the same line stays in float32 and torch then refuses the np.float32 assignment
      (graphics_utils.py:41).  The intrinsics are therefore handed over as a float64 tensor holding float32-rounded values:
      exactly the arithmetic of the reference's own environment.

  python tests/golden/make_golden_mg.py
"""
import math
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import ref_embed  # noqa: E402
import ref_import  # noqa: E402

OUT = os.path.join(HERE, "mg")


def embed_case(name, use_action, semantic, seed, d_hidden=32, N=129, D=6):
    net = ref_embed.build_net(d_hidden, use_action=use_action, semantic=semantic, seed=seed)
    assert net is not None, "needs /root/reference"
    data = ref_embed.make_inputs(N, D=D, seed=seed, use_action=use_action)
    r = ref_embed.run(net, data)
    out = dict(case=np.frombuffer(repr(dict(d_hidden=d_hidden, use_action=use_action, semantic=semantic, seed=seed, N=N,
                                            D=D)).encode(), dtype=np.uint8))
    for k in ("xyz", "dec_fts", "lang", "action"):
        if k in data:
            out["in_" + k] = data[k].numpy()
    for n_, p in net.state_dict().items():
        out["sd_" + n_] = p.numpy()
    for k in ref_embed.OUT_KEYS:
        out["out_" + k] = r.data[k].detach().numpy()
    for k in r.data["next"]:
        out["next_" + k] = r.data["next"][k].detach().numpy()
    for k, v in r.probe.items():
        out["probe_" + k] = v.detach().numpy()
    for k, v in r.cot.items():
        out["cot_" + k] = v.numpy()
    for k, v in r.grad.items():
        if k.startswith("param:") and not k.startswith("param:gs_deformation_field."):
            continue  # the encoder / regressor GEMMs are torch's on both sides: their parameter gradients pin nothing here
        out["grad_" + k] = v.numpy()
    os.makedirs(OUT, exist_ok=True)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
    print(name, "clamped scales:", float((r.data["scale_maps"] == 0.05).float().mean()),
          "points outside the volume:", int(((r.probe["latent_in"][:, 128:131] < 0) | (r.probe["latent_in"][:, 128:131] > 1)).any(1).sum()))


def cameras(V, W, H, neg, seed):
    from manigaussian_amd import synthetic as syn
    rng = np.random.default_rng(seed)
    c2w, K = [], []
    for v in range(V):
        th = 2 * math.pi * v / V
        eye = np.array([0.2 + 1.3 * math.cos(th), 1.3 * math.sin(th), 0.9 + rng.uniform(0.0, 0.6)])
        c2w.append(syn.look_at_c2w(eye, (0.2, 0.0, 0.9), flip_xy=neg))
        f = (W / 2) / math.tan(math.radians(rng.uniform(15.0, 35.0))) * (-1 if neg else 1)
        K.append(np.array([[f, 0, W / 2 + rng.uniform(-3, 3)], [0, f * rng.uniform(0.9, 1.1), H / 2 + rng.uniform(-3, 3)],
                           [0, 0, 1]]))
    return np.stack(c2w).astype(np.float32), np.stack(K).astype(np.float32)


def calib_cases():
    NR = ref_import.load_neural_rendering()
    assert NR is not None, "needs /root/reference"
    out = {}
    specs = [("neg_128", 128, 128, True, 0.1, 4.0, (0.0, 0.0, 0.0), 1.0), ("pos_128x96", 128, 96, False, 0.1, 4.0, (0.0, 0.0, 0.0), 1.0),
             ("neg_256_moved", 256, 256, True, 0.05, 10.0, (0.1, -0.2, 0.05), 1.5)]
    for i, (name, W, H, neg, zn, zf, trans, scale) in enumerate(specs):
        c2w, K = cameras(7, W, H, neg, seed=20 + i)
        self_ = types.SimpleNamespace(W=W, H=H, znear=zn, zfar=zf, trans=list(trans), scale=scale)
        data = dict(intr=torch.from_numpy(K.astype(np.float64)), extr=torch.from_numpy(c2w))
        nv = NR.NeuralRenderer.get_novel_calib(self_, data)
        out[name + "/spec"] = np.array([W, H, zn, zf, *trans, scale], np.float64)
        out[name + "/c2w"], out[name + "/K"] = c2w, K
        for k in ("FovX", "FovY", "world_view_transform", "full_proj_transform", "camera_center", "width", "height"):
            out[name + "/" + k] = nv[k].numpy()
        print(name, "FovX", nv["FovX"][:3].tolist())
    os.makedirs(OUT, exist_ok=True)
    np.savez_compressed(os.path.join(OUT, "novel_calib.npz"), **out)


if __name__ == "__main__":
    embed_case("embed_h32_action", use_action=True, semantic=False, seed=11)
    embed_case("embed_h32_semantic", use_action=False, semantic=True, seed=12)
    calib_cases()
