"""Phase timeline of the render backward at C3 (diagnostic): python scripts/trace_bwd.py"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from manigaussian_amd import GaussianRasterizationSettings, GaussianRasterizer, _lib
from manigaussian_amd import synthetic as syn

dev = torch.device("cuda:0")
P = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
F = int(sys.argv[2]) if len(sys.argv) > 2 else 32
W = 128
sc = {k: v.to(dev).requires_grad_(True) for k, v in syn.make_scene(P, F=F, M=4, seed=0).items()}
cam = syn.circle_cameras(8, W, W, negative_focal=True)[0]
rast = GaussianRasterizer(GaussianRasterizationSettings(**syn.camera_settings_kwargs(cam, 1, True, device=dev)))
dC, dF = [t.to(dev) for t in syn.make_cotangents(W, W, F)]
m2 = torch.zeros(P, 3, device=dev)


def step():
    c, f, r = rast(sc["means3D"], m2, sc["opacities"], shs=sc["shs"], language_feature_precomp=sc["language_feature"],
                   scales=sc["scales"], rotations=sc["rotations"])
    return torch.autograd.grad([c, f], list(sc.values()), [dC, dF])


for _ in range(5):
    step()
_lib.set_option("dbg", 256)
step()
torch.cuda.synchronize()
_lib.set_option("dbg", 0)
EV = 16
buf = np.zeros(512 * 16 * EV, np.uint64)
assert _lib.lib().mgs_debug_read_trace_bwd(buf.ctypes.data, buf.size) == 0
t = buf.reshape(512, 16, EV).astype(np.int64)[:256]
t0 = np.where(t[:, :, 0] > 0, t[:, :, 0], np.iinfo(np.int64).max).min(1)
rel = np.where(t > 0, t - t0[:, None, None], -1)
names = {0: "entry", 1: "dL loaded", 2: "q written", 3: "barrier", 10: "first chunk staged", 11: "barrier 2", 4: "chunk state (B, T_in)", 5: "records staged",
         6: "group 1 pixel loops", 7: "group 1 sums out", 8: "group 0 pixel loops", 9: "group 0 sums out", 15: "exit"}
print("shader-clock cycles since the block's first stamp; per block the LAST wave that stamped the event")
for e, n in names.items():
    v = rel[:, :, e].max(1)
    m = v >= 0
    if m.any():
        print(f"{n:24s} ev {e:2d}: blocks={int(m.sum()):4d} mean {v[m].mean():9.0f}  p50 {np.median(v[m]):9.0f}  max {v[m].max():8d}")
ends = rel[:, :, 15].max(1)
print("block duration: min", ends.min(), "mean", int(ends.mean()), "p90", int(np.percentile(ends, 90)), "max", ends.max())
# per-wave durations of the phases for waves that processed a chunk
live = rel[:, :, 9] >= 0
for a, b, n in ((4, 5, "staging"), (5, 6, "group-1 loops"), (6, 7, "group-1 out"), (7, 8, "group-0 loops"), (8, 9, "group-0 out")):
    ok = live & (rel[:, :, a] >= 0) & (rel[:, :, b] >= 0)
    d = (rel[:, :, b] - rel[:, :, a])[ok]
    if d.size:
        print(f"  per live wave {n:14s}: mean {d.mean():8.0f}  p90 {np.percentile(d, 90):8.0f}  max {d.max():8d}  (waves {d.size})")
print("live waves per block: mean", live.sum(1).mean(), "max", live.sum(1).max())
# per wave of the slowest blocks: cycles from "records staged" to "group 0 sums out" (the wave's whole chunk), by wave index
order = np.argsort(-ends)[:6]
for b in order:
    d = np.where((rel[b, :, 9] >= 0) & (rel[b, :, 4] >= 0), rel[b, :, 9] - rel[b, :, 4], -1)
    print(f"block {b}: duration {ends[b]}, live waves {int(live[b].sum())}, per wave (index = wave; chunks 4-7 sit on waves 7-4) chunk cycles {d[:12].tolist()}, "
          f"chunk end {rel[b, :12, 9].tolist()}")
med = np.argsort(ends)[len(ends) // 2]
d = np.where((rel[med, :, 9] >= 0) & (rel[med, :, 4] >= 0), rel[med, :, 9] - rel[med, :, 4], -1)
print(f"median block {med}: duration {ends[med]}, live waves {int(live[med].sum())}, per wave chunk cycles {d[:12].tolist()}")
