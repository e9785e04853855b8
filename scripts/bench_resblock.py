"""EXPERIMENT (round 5): one ResnetBlockFC of the deformation MLP (MG/resnetfc.py:10-62, d_hidden 512) as ONE hand-written kernel
(scripts/ubench/resblock_fused.hip -> libresblock.so, `make -C scripts/ubench`) against what the product runs for the same block -- hipBLASLt / rocBLAS GEMMs through torch + the streaming passes of csrc/mgs_mlp.hip
(manigaussian_amd/deform.py _FusedResnetFC).  Forward and data-gradient backward, fp32, correctness first, then hipEvent timing.

  python scripts/bench_resblock.py [M=100000] [iters=20] [tune=0]
"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from manigaussian_amd import _lib, deform

kv = dict(a.split("=") for a in sys.argv[1:])
M, iters, tune = int(kv.get("M", 100000)), int(kv.get("iters", 20)), int(kv.get("tune", 0))
H = 512
dev = torch.device("cuda:0")
L = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "ubench", "libresblock.so"))
c_fp = ctypes.c_void_p
L.mgs_mlp_pack_weight.argtypes = [c_fp, ctypes.c_int, c_fp, c_fp]
L.mgs_mlp_resblock_forward.argtypes = [ctypes.c_int] + [c_fp] * 9
L.mgs_mlp_resblock_backward.argtypes = [ctypes.c_int] + [c_fp] * 10
g_ = torch.Generator().manual_seed(0)
s = torch.randn(M, H, generator=g_).to(dev)
W0 = (torch.randn(H, H, generator=g_) / H ** 0.5).to(dev)
W1 = (torch.randn(H, H, generator=g_) / H ** 0.5).to(dev)
b0, b1 = torch.randn(H, generator=g_).mul(0.1).to(dev), torch.randn(H, generator=g_).mul(0.1).to(dev)
gout = torch.randn(M, H, generator=g_).to(dev)
if tune:
    deform.tune_gemms()
st = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)


def torch_fwd():
    a, xb = deform._relu_bias(s, b1)
    h = deform._addmm_relu(b0, a, W0.t())
    xb.addmm_(h, W1.t())
    return a, h, xb


def torch_bwd(h):
    db0, cs = torch.zeros(H, device=dev), torch.zeros(H, device=dev)
    gh = deform._relu_backward(gout @ W1, h, None, db0)
    g2 = deform._relu_backward(gh @ W0, s, gout, cs)
    return gh, g2, db0, cs


def pack(W, transpose):
    Wp = torch.empty(H * H, device=dev)
    _lib.check(L.mgs_mlp_pack_weight(W.data_ptr(), transpose, Wp.data_ptr(), st), "pack")
    return Wp


def fused_fwd(W0p, W1p, bufs=None):
    a, h, out = bufs or (torch.empty_like(s), torch.empty_like(s), torch.empty_like(s))
    _lib.check(L.mgs_mlp_resblock_forward(M, s.data_ptr(), W0p.data_ptr(), b0.data_ptr(), W1p.data_ptr(), b1.data_ptr(),
                                          a.data_ptr(), h.data_ptr(), out.data_ptr(), st), "fused fwd")
    return a, h, out


def fused_bwd(h, W1q, W0q, bufs=None):
    gh, g2 = bufs or (torch.empty_like(s), torch.empty_like(s))
    db0, cs = torch.zeros(H, device=dev), torch.zeros(H, device=dev)
    _lib.check(L.mgs_mlp_resblock_backward(M, gout.data_ptr(), h.data_ptr(), s.data_ptr(), W1q.data_ptr(), W0q.data_ptr(),
                                           gh.data_ptr(), g2.data_ptr(), db0.data_ptr(), cs.data_ptr(), st), "fused bwd")
    return gh, g2, db0, cs


def timed(fn, n):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def rel(x, y):
    return ((x - y).abs().max() / (y.abs().max() + 1e-30)).item()


for _ in range(3 if not tune else 12):
    ta, th, tout = torch_fwd()
    torch_bwd(th)
W0p, W1p = pack(W0, 1), pack(W1, 1)      # forward: x W^T
W1q, W0q = pack(W1, 0), pack(W0, 0)      # backward: g W
fa, fh, fout = fused_fwd(W0p, W1p)
print(f"forward  vs torch: a {rel(fa, ta):.2e}  h {rel(fh, th):.2e}  out {rel(fout, tout):.2e}   (relative to the tensor's max)")
tgh, tg2, tdb0, tcs = torch_bwd(th)
fgh, fg2, fdb0, fcs = fused_bwd(th, W1q, W0q)
print(f"backward vs torch: gh {rel(fgh, tgh):.2e}  g' {rel(fg2, tg2):.2e}  colsum(gh) {rel(fdb0, tdb0):.2e}  colsum(g') {rel(fcs, tcs):.2e}")
flops = 2 * 2 * M * H * H
bufs3 = (torch.empty_like(s), torch.empty_like(s), torch.empty_like(s))
bufs2 = (torch.empty_like(s), torch.empty_like(s))
t_tf = timed(torch_fwd, iters)
t_ff = timed(lambda: fused_fwd(W0p, W1p, bufs3), iters)
t_tb = timed(lambda: torch_bwd(th), iters)
t_fb = timed(lambda: fused_bwd(th, W1q, W0q, bufs2), iters)
t_pack = timed(lambda: pack(W0, 1), iters)
peak = 157.3
for name, t in (("torch  forward  (relu/bias pass + 2 GEMMs, bias+ReLU and beta=1 epilogues)", t_tf),
                ("fused  forward  (one kernel)", t_ff),
                ("torch  backward (2 GEMMs + 2 mask / residual / column-sum passes)", t_tb),
                ("fused  backward (one kernel)", t_fb)):
    print(f"M={M}  {name:80s} {t * 1e3:8.1f} us   {flops / (t * 1e-3) / 1e12:6.1f} TFLOP/s = {flops / (t * 1e-3) / 1e12 / peak:.3f} of the fp32 matrix peak")
print(f"weight repack (once per optimizer step and weight): {t_pack * 1e3:.1f} us")
print(f"library {_lib.build_id()}  tuned GEMMs: {bool(tune)}")
