"""Generates tests/golden/*.npz: seeded inputs + Oracle B outputs (images, radii, gradients).

These are NOT reference outputs (those live in tests/golden/ref, see make_golden_ref.py; the reference build
is fixed at 3 feature channels, these cover F = 32 as well): they freeze the oracle (itself pinned by the
reference goldens, the autograd Oracle A and closed forms in tests/test_oracle.py) so that (a) an accidental change of the oracle is caught on CPU and (b) the GPU parity
tests have committed vectors that do not depend on the oracle building on the GPU box.

  python tests/golden/make_golden.py
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))
import util  # noqa: E402

CASES = {
    "sh_f3_negfocal_48x40": dict(P=300, F=3, W=48, H=40, neg=True, bg=(0.1, 0.2, 0.3)),
    "precomp_f32_posfocal_32x32": dict(P=250, F=32, W=32, H=32, neg=False, colors_precomp=True, bg=(0.0, 0.0, 0.0)),
    "sh3_nofeat_unnorm_32x32": dict(P=200, F=3, M=16, sh_degree=3, W=32, H=32, include_feature=False,
                                    unnormalized_rot=True, bg=(0.5, 0.0, 0.25)),
}


def main():
    for name, c in CASES.items():
        sc, cam, kw, dC, dF = util.scene_case(**c)
        color, feat, radii, grads, state = util.run_oracle_b(sc, kw, dC, dF)
        out = {f"in_{k}": v.numpy() for k, v in sc.items()}
        out.update({f"cam_{k}": (v.numpy() if isinstance(v, torch.Tensor) else np.asarray(v)) for k, v in cam.items()})
        out["case"] = np.frombuffer(repr(c).encode(), dtype=np.uint8)
        out["d_color"] = dC.numpy()
        if dF is not None:
            out["d_feat"] = dF.numpy()
        out.update(out_color=color.numpy(), out_feat=feat.numpy(), radii=radii.numpy(),
                   num_rendered=np.int64(state.num_rendered))
        out.update({f"grad_{k}": v.numpy() for k, v in grads.items()})
        from oracle import oracle_b  # which pixels / Gaussians hold a pair within 2e-5 of a hard threshold (none, in these)
        out["fragile_pixels"] = oracle_b.fragile_mask(state).numpy().astype(np.uint8)
        out["fragile_gaussians"] = oracle_b.fragile_gaussians(state).numpy().astype(np.uint8)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
        print(name, "R =", state.num_rendered, "visible =", int((radii > 0).sum()))


if __name__ == "__main__":
    main()
