"""First-contact GPU diagnostic: self-test, parity vs Oracle B for a few configs and both backward
reduction variants, rough timings.  Run on the GPU box:  python scripts/gpu_check.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch

from manigaussian_amd import _lib
import util


def main():
    L = _lib.lib()
    print("device:", torch.cuda.get_device_name(0))
    rc = L.mgs_selftest(None)
    print("selftest rc =", rc, _lib.last_error() if rc else "")
    cases = [dict(P=3000, F=3), dict(P=3000, F=32), dict(P=3000, F=3, neg=False, colors_precomp=True),
             dict(P=20000, F=32), dict(P=2000, F=5), dict(P=2000, F=3, include_feature=False)]
    for red in (0, 1):
        _lib.set_option("bwd_reduce", red)
        for tb in (0, 1):
            _lib.set_option("tight_bins", tb)
            for c in cases:
                sc, cam, kw, dC, dF = util.scene_case(**c)
                inc = c.get("include_feature", True)
                cr, fr, rr, gr, st = util.run_oracle_b(sc, kw, dC, dF)
                t0 = time.time()
                ch, fh, rh, gh = util.run_hip(sc, cam, dC, dF, 1, inc, (0.1, 0.2, 0.3))
                dt = time.time() - t0
                ec = (ch - cr).abs().max().item()
                ef = (fh - fr).abs().max().item() if inc else 0.0
                errs = util.grad_errors(gh, gr)
                worst = max((e / (m + 1e-30), k) for k, (e, m) in errs.items() if m > 0)
                print(f"red={red} tight={tb} {c}: R={st.num_rendered} radii_eq={bool((rh == rr).all())} "
                      f"color {ec:.2e} feat {ef:.2e} worst grad rel-to-max {worst[0]:.2e} ({worst[1]}) [{dt*1e3:.0f} ms]")
                if worst[0] > 1e-3 or ec > 1e-4:
                    for k, (e, m) in errs.items():
                        print(f"      {k}: err {e:.3e} max {m:.3e}")
    # timing at the headline config
    _lib.set_option("bwd_reduce", 1)
    for tb in (0, 1):
        _lib.set_option("tight_bins", tb)
        sc, cam, kw, dC, dF = util.scene_case(P=100000, F=32)
        util.run_hip(sc, cam, dC, dF, 1, True, (0.0, 0.0, 0.0))
        torch.cuda.synchronize()
        t0 = time.time()
        for _ in range(5):
            util.run_hip(sc, cam, dC, dF, 1, True, (0.0, 0.0, 0.0))
        torch.cuda.synchronize()
        print(f"tight={tb}: P=100k F=32 fwd+bwd incl. host overhead+H2D: {(time.time() - t0) / 5 * 1e3:.2f} ms")


if __name__ == "__main__":
    main()
