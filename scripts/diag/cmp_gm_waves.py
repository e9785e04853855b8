import sys, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import util
from manigaussian_amd import _lib, _C, GaussianRasterizationSettings, GaussianRasterizer
from manigaussian_amd import synthetic as syn
dev = torch.device("cuda:0")
for P, F in ((100000, 32), (30000, 3)):
    sc, cam, kw, dC, dF = util.scene_case(P=P, F=F)
    res = {}
    for gw in (16, 12, 13, 14):
        _lib.set_option("gm_waves", gw)
        res[gw] = util.run_hip(sc, cam, dC, dF, 1, True, (0.1, 0.2, 0.3))
    _lib.set_option("gm_waves", 16)
    for gw in (12, 13, 14):
        worst = 0.0
        for k, v in res[16][3].items():
            d = (res[gw][3][k] - v).abs().max().item() / (v.abs().max().item() + 1e-30)
            worst = max(worst, d)
        print(f"P={P} F={F} gm_waves={gw}: images equal {torch.equal(res[gw][0], res[16][0])}, worst gradient difference vs 16 waves {worst:.2e} of the tensor max")
