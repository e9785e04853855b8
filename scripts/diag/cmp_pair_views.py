"""Why does the two-pixel render backward (gm_waves 12) differ from the one-pixel forms by more than summation noise at
some Gaussians of tests/test_gpu_parity.py::test_view_batch_equals_per_view_calls[f32_8views]?  Per view: means2D gradient of
the 12-wave, 16-wave and 8-wave forms and of the reference's own kernels (run twice: its own atomic-order noise), at the
element where 12 and 16 differ most."""
import sys, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import util
from manigaussian_amd import _lib
from manigaussian_amd import synthetic as syn
from oracle import ref_cuda

P, F, V, W, H = 20000, 32, 8, 128, 128
bg = (0.1, 0.2, 0.3)
sc = syn.make_scene(P, F=F, M=4, seed=2)
cams = syn.circle_cameras(V, W, H, negative_focal=True)
g = torch.Generator().manual_seed(4)
dC, dF = torch.randn(V, 3, H, W, generator=g), torch.randn(V, F, H, W, generator=g)
have_ref = ref_cuda.available(F)
for v, cam in enumerate(cams):
    res = {}
    for gw in (16, 12, 8):
        _lib.set_option("gm_waves", gw)
        res[gw] = util.run_hip(sc, cam, dC[v], dF[v], 1, True, bg)[3]
    _lib.set_option("gm_waves", 12)
    refs = []
    if have_ref:
        kw = syn.camera_settings_kwargs(cam, 1, True, bg=bg)
        refs = [util.run_reference(sc, kw, dC[v], dF[v])[3] for _ in range(2)]
    for k in ("means2D", "opacities", "scales", "language_feature"):
        a, b, c = res[16][k], res[12][k], res[8][k]
        d = (a - b).abs()
        i = int(d.reshape(-1).argmax())
        mag = a.abs().max().item()
        line = (f"view {v} {k:16s} max|g| {mag:10.4g}  |12-16| {d.max().item():9.3g} ({d.max().item() / mag:8.2e})  "
                f"|8-16| {(a - c).abs().max().item() / mag:8.2e}  at {i}: 16 {a.reshape(-1)[i].item():+.6e} "
                f"12 {b.reshape(-1)[i].item():+.6e} 8 {c.reshape(-1)[i].item():+.6e}")
        if refs:
            r0 = refs[0][util.GRAD_KEYS[k]].reshape(a.shape)
            r1 = refs[1][util.GRAD_KEYS[k]].reshape(a.shape)
            line += (f" ref {r0.reshape(-1)[i].item():+.6e} / {r1.reshape(-1)[i].item():+.6e}; worst vs ref: "
                     f"16 {(a - r0).abs().max().item() / mag:8.2e} 12 {(b - r0).abs().max().item() / mag:8.2e} "
                     f"ref-ref {(r0 - r1).abs().max().item() / mag:8.2e}")
        print(line, flush=True)
