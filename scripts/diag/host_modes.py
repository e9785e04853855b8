"""Where the host time of the package-default path (forward mode "safe") goes against "async" under torch's default autograd
threading: the binding's own segment clocks, the allocator's counters and the wall time of the two halves of a step."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import manigaussian_amd as mg
from manigaussian_amd import GaussianRasterizationSettings, GaussianRasterizer, check_status, _C
from manigaussian_amd import synthetic as syn

P, F, W = int(os.environ.get("HP_P", "1000")), int(os.environ.get("HP_F", "32")), 128
dev = torch.device("cuda:0")
sc = syn.make_scene(P, F=F, M=4, seed=0)
cam = syn.circle_cameras(8, W, W, negative_focal=True)[0]
params = {k: v.to(dev).requires_grad_(True) for k, v in sc.items()}
plist = list(params.values())
rast = GaussianRasterizer(GaussianRasterizationSettings(**syn.camera_settings_kwargs(cam, 1, True, device=dev)))
dC, dF = [t.to(dev) for t in syn.make_cotangents(W, W, F)]
m2 = torch.zeros(P, 3, device=dev, requires_grad=True)
e = _C.compiled()
N = 2000
tf = tb = 0.0


def step():
    global tf, tb
    t0 = time.perf_counter()
    c, f, r = rast(means3D=params["means3D"], means2D=m2, opacities=params["opacities"], shs=params["shs"],
                   language_feature_precomp=params["language_feature"], scales=params["scales"], rotations=params["rotations"])
    t1 = time.perf_counter()
    g = torch.autograd.grad([c, f], plist, [dC, dF])
    t2 = time.perf_counter()
    tf += t1 - t0
    tb += t2 - t1
    return g


for fmode in ("safe", "async", "safe", "async"):
    mg.set_forward_mode(fmode)
    for mt in (True, False):
        torch.autograd.set_multithreading_enabled(mt)
        for _ in range(100):
            step()
        check_status(dev)
        torch.cuda.synchronize()
        s0 = torch.cuda.memory_stats(dev)
        e.set_profile(True)
        e.profile_read(True)
        tf = tb = 0.0
        t0 = time.perf_counter()
        for _ in range(N):
            step()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        e.set_profile(False)
        prof = e.profile_read(True)
        s1 = torch.cuda.memory_stats(dev)
        print(f"P={P} mode={fmode} multithreading={mt}: host {(t1 - t0) / N * 1e6:.1f} us / step ({(t2 - t0) / N * 1e6:.1f} incl. drain); "
              f"forward call {tf / N * 1e6:.1f}, autograd.grad {tb / N * 1e6:.1f}")
        print("   segments:", ", ".join(f"{k.split('.', 1)[1] if k.startswith('fwd') else k}={ns / n / 1e3:.2f}" for k, (ns, n) in prof.items() if n))
        print("   allocator: " + ", ".join(f"{k}={s1[k] - s0[k]}" for k in ("num_device_alloc", "num_device_free", "num_alloc_retries",
              "allocation.all.allocated", "segment.all.allocated") if k in s1),
              f"reserved={s1['reserved_bytes.all.current'] / 2**20:.0f} MB, ws held={e.held_bytes(0) if hasattr(e, 'held_bytes') else None}")
torch.autograd.set_multithreading_enabled(True)
print("counters:", e.counters())
