"""Which option decides the one pixel of the 500 000-Gaussian / 256 x 256 live-reference case that sits above 1e-4?
Runs the HIP path under several MgsOptions against the reference kernels (oracle/_ref) and prints, per variant, the worst
pixel of the colour and feature images."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import util
from manigaussian_amd import _lib

case = dict(P=500000, F=32, W=256, H=256, neg=True, bg=(0.0, 0.0, 0.0), seed=0)
sc, cam, kw, dC, dF = util.scene_case(**case)
cr, fr, rr, gr, R = util.run_reference(sc, kw, dC, dF)
base = dict(tight_bins=1, fast_exp=1, exact_cull=1)
for name, ov in [("default", {}), ("tight_bins=0", dict(tight_bins=0)), ("exact_cull=0", dict(exact_cull=0)),
                 ("tight=0 cull=0", dict(tight_bins=0, exact_cull=0)), ("all exact", dict(tight_bins=0, exact_cull=0, fast_exp=0)),
                 ("gm_waves=8", dict(gm_waves=8))]:
    for k, v in {**base, **ov}.items():
        _lib.set_option(k, v)
    ch, fh, rh, gh = util.run_hip(sc, cam, dC, dF, 1, True, case["bg"])
    ec = np.abs(ch.numpy() - np.asarray(cr)).max(0); ef = np.abs(fh.numpy() - np.asarray(fr)).max(0)
    yc, xc = np.unravel_index(ec.argmax(), ec.shape); yf, xf = np.unravel_index(ef.argmax(), ef.shape)
    print(f"{name:16s} colour max {ec.max():.3e} at (y={yc}, x={xc}) n>2e-5: {(ec > 2e-5).sum()}  feature max {ef.max():.3e} at (y={yf}, x={xf}) "
          f"n>2e-5: {(ef > 2e-5).sum()}  colour hip/ref at pixel: {ch[:, yc, xc].tolist()} / {np.asarray(cr)[:, yc, xc].tolist()}")
for k, v in base.items():
    _lib.set_option(k, v)
_lib.set_option("gm_waves", 16)
