import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, util
from oracle import oracle_b
for case in [dict(P=6000, F=32, neg=False), dict(P=3000, F=3, M=16, sh_degree=3, unnormalized_rot=True)]:
    sc, cam, kw, dC, dF = util.scene_case(**case)
    cr, fr, rr, gr, st = util.run_oracle_b(sc, kw, dC, dF)
    ch, fh, rh, gh = util.run_hip(sc, cam, dC, dF, case.get("sh_degree", 1), True, (0.1, 0.2, 0.3))
    for eps in (2e-5, 1e-4):
        fg = oracle_b.fragile_gaussians(st, eps); fm = oracle_b.fragile_mask(st, eps)
        print(case, 'eps', eps, 'fragile px frac', fm.float().mean().item(), 'fragile gauss frac', fg.float().mean().item())
        e = (ch - cr).abs().max(0)[0]
        print('  color err robust', e[~fm].max().item(), 'fragile', e[fm].max().item() if fm.any() else 0)
        for k in ['means3D', 'scales', 'rotations', 'opacities', 'shs', 'language_feature']:
            v = gh[k]; r = gr[util.GRAD_KEYS[k]].reshape(v.shape)
            d = (v - r).abs().reshape(v.shape[0], -1).max(1)[0]
            m = r.abs().max().item()
            print(f'  {k}: robust {d[~fg].max().item()/m:.2e} fragile {(d[fg].max().item()/m if fg.any() else 0):.2e}  n_bad={(d > 1e-4*m).sum().item()} of which fragile {((d > 1e-4*m) & fg).sum().item()}')
