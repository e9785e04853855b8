"""Host cost of one training step through the PUBLIC autograd API (small P: the GPU is never the bottleneck), round 5:
the compiled binding (csrc/mgs_torch.cpp) against the ctypes shim (_C.py), package-default options ("safe") and "async", torch's
default autograd threading and set_multithreading_enabled(False); then where the compiled path's time goes (cProfile sees the
Python side only: one call into the binding per forward, one run_backward per step) and the library's own cost per call
(mgs_rasterize_forward / _backward called in a loop through ctypes: launches + argument checks, no torch)."""
import cProfile, ctypes, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import manigaussian_amd as mg
from manigaussian_amd import GaussianRasterizationSettings, GaussianRasterizer, check_status, _C, _lib
from manigaussian_amd import synthetic as syn

P, F, W = int(os.environ.get("HP_P", "1000")), 32, 128
dev = torch.device("cuda:0")
sc = syn.make_scene(P, F=F, M=4, seed=0)
cam = syn.circle_cameras(8, W, W, negative_focal=True)[0]
params = {k: v.to(dev).requires_grad_(True) for k, v in sc.items()}
plist = list(params.values())
rast = GaussianRasterizer(GaussianRasterizationSettings(**syn.camera_settings_kwargs(cam, 1, True, device=dev)))
dC, dF = [t.to(dev) for t in syn.make_cotangents(W, W, F)]
m2 = torch.zeros(P, 3, device=dev, requires_grad=True)


def step():
    c, f, r = rast(means3D=params["means3D"], means2D=m2, opacities=params["opacities"], shs=params["shs"],
                   language_feature_precomp=params["language_feature"], scales=params["scales"], rotations=params["rotations"])
    return torch.autograd.grad([c, f], plist, [dC, dF])


def fwd_only():
    with torch.no_grad():
        return rast(means3D=params["means3D"], means2D=m2, opacities=params["opacities"], shs=params["shs"],
                    language_feature_precomp=params["language_feature"], scales=params["scales"], rotations=params["rotations"])


N = 2000
print(f"compiled binding present: {_C.compiled() is not None}; library {_lib.build_id()}")
for compiled in (True, False):
    for fmode in ("safe", "async"):
        mg.set_forward_mode(fmode)
        with _C.use_compiled(compiled):
            for _ in range(50):
                step()
            check_status(dev)
            for name, mt in (("default autograd threading", True), ("set_multithreading_enabled(False)", False)):
                torch.autograd.set_multithreading_enabled(mt)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(N):
                    step()
                t1 = time.perf_counter()
                torch.cuda.synchronize()
                t2 = time.perf_counter()
                print(f"P={P} {'compiled' if compiled else 'ctypes  '} forward_mode={fmode:5s} {name}: "
                      f"{(t1 - t0) / N * 1e6:6.1f} us of host time per fwd+bwd (enqueue only), {(t2 - t0) / N * 1e6:6.1f} us incl. drain")
            torch.autograd.set_multithreading_enabled(False)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(N):
                fwd_only()
            t1 = time.perf_counter()
            torch.cuda.synchronize()
            print(f"P={P} {'compiled' if compiled else 'ctypes  '} forward_mode={fmode:5s} forward only (no_grad): {(t1 - t0) / N * 1e6:6.1f} us")
            check_status(dev)
torch.autograd.set_multithreading_enabled(False)
mg.set_forward_mode("safe")
e = _C.compiled()
if e is not None:
    print("binding counters:", e.counters())
if e is not None:  # where the binding's own time goes (steady_clock laps inside rasterize() / the node's apply())
    e.set_profile(True)
    e.profile_read(True)
    for _ in range(N):
        step()
    torch.cuda.synchronize()
    e.set_profile(False)
    prof = e.profile_read(True)
    print("---- the compiled binding, segment by segment (us per call) ----")
    for k, (ns, n) in prof.items():
        if n:
            print(f"  {k:28s} {ns / n / 1e3:7.2f}")
    print(f"  {'sum fwd / bwd':28s} {sum(ns / n for k, (ns, n) in prof.items() if n and k.startswith('fwd')) / 1e3:7.2f} / "
          f"{sum(ns / n for k, (ns, n) in prof.items() if n and k.startswith('bwd')) / 1e3:7.2f}")
pr = cProfile.Profile()
pr.enable()
for _ in range(N):
    step()
pr.disable()
torch.cuda.synchronize()
print("---- cProfile of the compiled path (package defaults), Python side ----")
pstats.Stats(pr).sort_stats("tottime").print_stats(12)

# ---- the library alone: forward + backward through the C ABI in a loop (what no binding can go below) ----
with _C.use_compiled(False):
    mg.set_forward_mode("async")
    for _ in range(5):
        c, f, r = rast(means3D=params["means3D"], means2D=m2, opacities=params["opacities"], shs=params["shs"],
                       language_feature_precomp=params["language_feature"], scales=params["scales"], rotations=params["rotations"])
        check_status(dev)
    h = c.grad_fn.num_rendered
    L = _lib.lib()
    a = h.a
    st = mg._state.device_state(dev)
    out_c, out_f = torch.empty(3, W, W, device=dev), torch.empty(F, W, W, device=dev)
    radii = torch.empty(P, dtype=torch.int32, device=dev)
    nr = ctypes.c_int32(0)
    stream = _C._stream(dev)
    offs, total = _C._grad_offsets(L, P, 4, F)
    flat = torch.empty(total, device=dev)
    base = flat.data_ptr()
    (o_scr, o_col, o_feat, o_m3, o_op, o_sh, o_sc, o_rot, o_cov, o_m2, _p) = offs
    a.bwd_accum, a.bwd_accum_bytes, a.accum_prezeroed = None, 0, 0
    torch.cuda.synchronize()
    tf = tb = 0.0
    for i in range(N):
        slot, tag = st.take_slot()
        a.status_tag = tag
        t0 = time.perf_counter()
        rc = L.mgs_rasterize_forward(ctypes.byref(a), radii.data_ptr(), out_c.data_ptr(), out_f.data_ptr(), ctypes.byref(nr), slot, stream)
        t1 = time.perf_counter()
        rc2 = L.mgs_rasterize_backward(ctypes.byref(a), -1, radii.data_ptr(), dC.data_ptr(), dF.data_ptr(), base + 4 * o_m2, None,
                                       base + 4 * o_op, base + 4 * o_col, base + 4 * o_feat, base + 4 * o_m3, base + 4 * o_cov,
                                       base + 4 * o_sh, base + 4 * o_sc, base + 4 * o_rot, base + 4 * o_scr, (o_col - o_scr) * 4, stream)
        t2 = time.perf_counter()
        assert rc == 0 and rc2 == 0, (rc, rc2, _lib.last_error())
        tf += t1 - t0
        tb += t2 - t1
        if i % 200 == 199:
            torch.cuda.synchronize()
    torch.cuda.synchronize()
    print(f"library alone through ctypes: mgs_rasterize_forward {tf / N * 1e6:.1f} us (3 launches: preprocess with the keys, bucket rank, render; 5 in round 5), mgs_rasterize_backward "
          f"{tb / N * 1e6:.1f} us (3 launches incl. the accumulator fill) of host time per call")
