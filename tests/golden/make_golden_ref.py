"""Generates tests/golden/ref/*.npz: seeded inputs + the REFERENCE rasterizer's own outputs.

The outputs come from oracle/_ref/libmgs_ref.so (oracle/ref_cuda.py): the reference's forward.cu / backward.cu /
rasterizer_impl.cu compiled unmodified for gfx950.  It needs a GPU, so it runs on the GPU box; nothing here reads
/root/reference at run time (the .so is prebuilt and travels with the snapshot):

  gpurun -- 'python tests/golden/make_golden_ref.py gpurun_out/golden_ref'      # then copy into tests/golden/ref/

The stock reference build has F = 3 feature channels (RAST/cuda_rasterizer/config.h:15-16); the F = 32 and F = 8 cases come
from the same three sources rebuilt at that width (oracle/Makefile, REF_WIDTH: config.h's macros given on the command line) --
BASELINE configs[2]'s width pinned by COMMITTED vectors, not only by the live test that needs the prebuilt library.  The
scale_modifier cases exercise computeCov3D's `mod` (forward.cu:122-126) and the cov3D backward, whose dL_dscale deliberately
omits it (backward.cu:295,325-327).  Inputs are regenerated from `case` by tests/util.scene_case, and stored as well so a
drift of the generator is caught.  `ONLY=name1,name2` regenerates a subset.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))
import util  # noqa: E402

CASES = {
    "ref_sh_f3_negfocal_48x40": dict(P=300, F=3, W=48, H=40, neg=True, bg=(0.1, 0.2, 0.3)),
    "ref_precomp_f3_posfocal_32x32": dict(P=250, F=3, W=32, H=32, neg=False, colors_precomp=True, bg=(0.0, 0.0, 0.0)),
    "ref_sh3_nofeat_unnorm_32x32": dict(P=200, F=3, M=16, sh_degree=3, W=32, H=32, include_feature=False,
                                        unnormalized_rot=True, bg=(0.5, 0.0, 0.25)),
    "ref_cov3d_f3_64x64": dict(P=600, F=3, W=64, H=64, neg=True, cov3d=True, bg=(0.0, 0.0, 0.0)),
    "ref_sh2_f3_ragged_100x52": dict(P=1500, F=3, M=9, sh_degree=2, W=100, H=52, neg=True, bg=(1.0, 1.0, 1.0), seed=3,
                                     cam_index=2),
    "ref_sh_f3_128x128_p4000": dict(P=4000, F=3, W=128, H=128, neg=True, bg=(0.1, 0.2, 0.3), seed=7, cam_index=0),
    # round 5: other feature widths (the reference rebuilt), scale_modifier != 1
    "ref_sh_f32_negfocal_64x64": dict(P=900, F=32, W=64, H=64, neg=True, bg=(0.1, 0.2, 0.3), seed=21),
    "ref_sh2_f32_ragged_80x48": dict(P=1200, F=32, M=9, sh_degree=2, W=80, H=48, neg=False, bg=(0.0, 0.0, 0.0), seed=22,
                                     cam_index=2),
    "ref_sh_f8_48x48": dict(P=500, F=8, W=48, H=48, neg=True, bg=(0.2, 0.0, 0.1), seed=23, cam_index=0),
    "ref_scalemod0p5_f3_64x64": dict(P=800, F=3, W=64, H=64, neg=True, bg=(0.1, 0.2, 0.3), seed=24, scale_modifier=0.5),
    "ref_scalemod2_f32_64x64": dict(P=700, F=32, W=64, H=64, neg=True, bg=(0.0, 0.0, 0.0), seed=25, scale_modifier=2.0),
}


run_reference = util.run_reference


def main():
    out_dir = sys.argv[1] if len(sys.argv) > 1 else os.path.join(HERE, "ref")
    os.makedirs(out_dir, exist_ok=True)
    only = [n for n in os.environ.get("ONLY", "").split(",") if n]
    for name, c in CASES.items():
        if only and name not in only:
            continue
        sc, cam, kw, dC, dF = util.scene_case(**c)
        color, feat, radii, grads, R = run_reference(sc, kw, dC, dF)
        # atomics make the reference's gradients run-to-run non-deterministic at the ulp level: record the spread
        color2, feat2, radii2, grads2, R2 = run_reference(sc, kw, dC, dF)
        assert R2 == R and torch.equal(radii, radii2) and torch.equal(color, color2) and torch.equal(feat, feat2)
        spread = max(((grads[k] - grads2[k]).abs().max() / (grads[k].abs().max() + 1e-30)).item()
                     for k in grads if grads[k].numel())
        out = {f"in_{k}": v.numpy() for k, v in sc.items()}
        out["case"] = np.frombuffer(repr(c).encode(), dtype=np.uint8)
        out["d_color"] = dC.numpy()
        if dF is not None:
            out["d_feat"] = dF.numpy()
        out.update(out_color=color.numpy(), out_feat=feat.numpy(), radii=radii.numpy(), num_rendered=np.int64(R),
                   grad_spread=np.float64(spread))
        out.update({f"grad_{k}": v.numpy() for k, v in grads.items()})
        np.savez_compressed(os.path.join(out_dir, name + ".npz"), **out)
        print(f"{name}: R = {R}, visible = {int((radii > 0).sum())}, |color| max = {color.abs().max():.4f}, "
              f"grad run-to-run spread = {spread:.2e}", flush=True)


if __name__ == "__main__":
    main()
