"""GPU diagnostic: self-test, parity vs Oracle B across kernel variants, per-stage timings at the headline
config.  Run on the GPU box:  python tests/tools/gpu_check.py [quick]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from manigaussian_amd import _lib
import util


def parity(variants, cases):
    for v in variants:
        for k, val in v.items():
            _lib.set_option(k, val)
        for c in cases:
            sc, cam, kw, dC, dF = util.scene_case(**c)
            inc = c.get("include_feature", True)
            cr, fr, rr, gr, st = util.run_oracle_b(sc, kw, dC, dF)
            ch, fh, rh, gh = util.run_hip(sc, cam, dC, dF, c.get("sh_degree", 1), inc, c.get("bg", (0.1, 0.2, 0.3)))
            ec = util.image_errors(ch, cr, st)
            ef = util.image_errors(fh, fr, st) if inc else (0, 0, 0)
            errs, frac = util.grad_errors_split(gh, gr, st)
            wr = max((r / (m + 1e-30), k) for k, (r, f, m) in errs.items() if m > 0)
            wf = max((f / (m + 1e-30), k) for k, (r, f, m) in errs.items() if m > 0)
            ok = ec[0] <= 1e-4 and ef[0] <= 1e-4 and wr[0] <= 1e-3 and wf[0] <= 1e-2 and bool((rh == rr).all())
            print(f"{'OK ' if ok else 'BAD'} {v} {c}: R={st.num_rendered} color {ec[0]:.1e}/{ec[1]:.1e} "
                  f"feat {ef[0]:.1e}/{ef[1]:.1e} grad robust {wr[0]:.1e} ({wr[1]}) fragile {wf[0]:.1e} ({wf[1]})")


def timing(variants, P=100000, F=32, W=128, steps=30):
    from manigaussian_amd import GaussianRasterizationSettings, GaussianRasterizer
    from manigaussian_amd import synthetic as syn
    dev = torch.device("cuda:0")
    sc = syn.make_scene(P, F=F, M=4, seed=0)
    cam = syn.circle_cameras(8, W, W, negative_focal=True)[0]
    d = {k: v.to(dev).requires_grad_(True) for k, v in sc.items()}
    m2d = torch.zeros(P, 3, device=dev, requires_grad=True)
    dC, dF = [t.to(dev) for t in syn.make_cotangents(W, W, F)]
    rast = GaussianRasterizer(GaussianRasterizationSettings(**syn.camera_settings_kwargs(cam, 1, True, device=dev)))

    def step():
        c, f, r = rast(d["means3D"], m2d, d["opacities"], shs=d["shs"], language_feature_precomp=d["language_feature"],
                       scales=d["scales"], rotations=d["rotations"])
        torch.autograd.backward([c, f], [dC, dF])
        for t in d.values():
            t.grad = None
        m2d.grad = None

    for v in variants:
        for k, val in v.items():
            _lib.set_option(k, val)
        for _ in range(5):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) / steps * 1e3
        _lib.profile_read(True)
        _lib.set_option("profile", 2)
        for _ in range(10):
            step()
        torch.cuda.synchronize()
        _lib.set_option("profile", 0)
        st = {k: ms / max(c, 1) * 1e3 for k, (ms, c) in _lib.profile_read(True).items()}
        print(f"{v}: wall {wall:.3f} ms/step | " + " ".join(f"{k}={x:.0f}us" for k, x in st.items()) +
              f" | sum={sum(st.values()):.0f}us")


def _parse_variants(args):
    """'a=1,b=2 a=3' -> [{a:1,b:2},{a:3}]"""
    out = []
    for a in args:
        out.append({kv.split("=")[0]: int(kv.split("=")[1]) for kv in a.split(",") if kv})
    return out


def main():
    """gpu_check.py [quick] | parity <variants...> | timing <variants...>   (variant = k=v,k=v)"""
    L = _lib.lib()
    print("device:", torch.cuda.get_device_name(0))
    rc = L.mgs_selftest(None)
    print("selftest rc =", rc, _lib.last_error() if rc else "")
    cases = [dict(P=3000, F=3), dict(P=3000, F=32), dict(P=3000, F=3, neg=False, colors_precomp=True),
             dict(P=20000, F=32), dict(P=2000, F=5), dict(P=2000, F=3, include_feature=False),
             dict(P=3000, F=3, cov3d=True, W=72, H=40), dict(P=30000, F=32, W=64, H=64)]
    big = [dict(P=1500, F=64), dict(P=40000, F=3, W=256, H=256)]
    if len(sys.argv) > 2 and sys.argv[1] in ("parity", "timing"):
        vs = _parse_variants(sys.argv[2:])
        if sys.argv[1] == "parity":
            parity(vs, cases + big)
        else:
            timing(vs)
        return
    quick = len(sys.argv) > 1 and sys.argv[1] == "quick"
    variants = [dict(render_mode=2, chunk=128, tight_bins=1, exact_cull=1, fast_exp=0, bwd_reduce=1),
                dict(render_mode=2, chunk=64, tight_bins=0, exact_cull=0, fast_exp=1),
                dict(render_mode=2, chunk=256, tight_bins=1, exact_cull=1, fast_exp=1, bwd_reduce=0)]
    parity(variants[:1] if quick else variants, cases + big)
    _lib.set_option("bwd_reduce", 1)
    base = dict(tight_bins=1, fast_exp=0, exact_cull=1)
    timing([dict(render_mode=0, **base), dict(render_mode=1, chunk=128, **base),
            dict(render_mode=2, chunk=64, **base), dict(render_mode=2, chunk=128, **base),
            dict(render_mode=2, chunk=256, **base), dict(render_mode=2, chunk=128, tight_bins=1, fast_exp=1, exact_cull=1),
            dict(render_mode=2, chunk=128, tight_bins=1, fast_exp=1, exact_cull=0),
            dict(render_mode=2, chunk=128, tight_bins=0, fast_exp=1, exact_cull=1)])


if __name__ == "__main__":
    main()
