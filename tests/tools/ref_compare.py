"""Three-way comparison on the GPU box: reference kernels (oracle/_ref) vs Oracle B (CPU) vs the HIP path.

  python tests/tools/ref_compare.py            # prints per-case errors; test infrastructure, not product
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import util  # noqa: E402
import make_golden_ref as mg  # noqa: E402

EXTRA = {
    "c3like_p20000_128": dict(P=20000, F=3, W=128, H=128, neg=True, bg=(0.1, 0.2, 0.3), seed=11),
    "c3_p100000_128": dict(P=100000, F=3, W=128, H=128, neg=True, bg=(0.0, 0.0, 0.0), seed=12),
}


def main():
    cases = dict(mg.CASES)
    cases.update(EXTRA)
    for name, c in cases.items():
        sc, cam, kw, dC, dF = util.scene_case(**c)
        inc = c.get("include_feature", True)
        cr, fr, rr, gr, R = mg.run_reference(sc, kw, dC, dF)
        cb, fb, rb, gb, st = util.run_oracle_b(sc, kw, dC, dF)
        ch, fh, rh, gh = util.run_hip(sc, cam, dC, dF, c.get("sh_degree", 1), inc, c.get("bg", (0.1, 0.2, 0.3)))
        print(f"== {name}: R_ref = {R}, R_oracle = {st.num_rendered}, radii ref==oracle {torch.equal(rr, rb)}, "
              f"ref==hip {torch.equal(rr, rh)}")
        for tag, (c2, f2, g2) in {"oracleB": (cb, fb, gb), "hip": (ch, fh, None)}.items():
            rob, frag, frac = util.image_errors(c2, cr, st)
            s = f"   {tag:8s} color robust {rob:.2e} fragile {frag:.2e} (fragile px {frac:.4f})"
            if inc:
                rob, frag, _ = util.image_errors(f2, fr, st)
                s += f"  feat robust {rob:.2e} fragile {frag:.2e}"
            print(s)
        # gradients: oracle-B names on both sides
        gh_named = {util.GRAD_KEYS[k]: v for k, v in gh.items()}
        from oracle import oracle_b
        fg = oracle_b.fragile_gaussians(st, 2e-5)
        for k in sorted(gr):
            ref = gr[k]
            if ref.numel() == 0 or k not in gb:
                continue
            mx = ref.abs().max().item()
            row = f"   grad {k:17s} |ref| {mx:.3e}"
            for tag, g2 in (("oracleB", gb), ("hip", gh_named)):
                if k not in g2 or g2[k].numel() != ref.numel():
                    continue
                d = (g2[k].reshape(ref.shape) - ref).abs().reshape(ref.shape[0], -1).max(1)[0]
                rob = d[~fg].max().item() if (~fg).any() else 0.0
                fr_ = d[fg].max().item() if fg.any() else 0.0
                row += f"  {tag}: robust {rob / (mx + 1e-30):.2e} fragile {fr_ / (mx + 1e-30):.2e}"
            print(row)
        print(f"   fragile gaussians {fg.float().mean().item():.4f}", flush=True)


if __name__ == "__main__":
    main()
