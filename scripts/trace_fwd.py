"""Phase timeline of the render forward at C3 (diagnostic): python scripts/trace_fwd.py"""
import ctypes
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
sc = {k: v.to(dev) for k, v in syn.make_scene(P, F=F, M=4, seed=0).items()}
cam = syn.circle_cameras(8, W, W, negative_focal=True)[0]
rast = GaussianRasterizer(GaussianRasterizationSettings(**syn.camera_settings_kwargs(cam, 1, True, device=dev)))


def fwd():
    with torch.no_grad():
        return rast(sc["means3D"], torch.zeros(P, 3, device=dev), sc["opacities"], shs=sc["shs"],
                    language_feature_precomp=sc["language_feature"], scales=sc["scales"], rotations=sc["rotations"])


for _ in range(5):
    fwd()
_lib.set_option("dbg", 256)
fwd()
torch.cuda.synchronize()
_lib.set_option("dbg", 0)
L = _lib.lib()
EV = 24
buf = np.zeros(512 * 16 * EV, np.uint64)
rc = L.mgs_debug_read_trace(buf.ctypes.data, buf.size)
assert rc == 0, rc
t = buf.reshape(512, 16, EV).astype(np.int64)[:256]
# every XCD has its own counter: times are taken relative to each block's own first stamp
t0 = np.where(t[:, :, 0] > 0, t[:, :, 0], np.iinfo(np.int64).max).min(1)
rel = np.where(t > 0, t - t0[:, None, None], -1)
names = {0: "entry", 1: "fill done r0", 2: "list barrier r0", 3: "rows staged r0", 4: "phase A done r0", 5: "Tp barrier r0",
         6: "phase B done r0", 7: "round end r0", 9: "fill done r1", 10: "list barrier r1", 11: "rows r1", 12: "phase A r1",
         13: "Tp barrier r1", 14: "phase B r1", 15: "round end r1", 21: "before final", 22: "image summed", 23: "exit"}
print(f"shader-clock cycles (s_memtime) since the block's first stamp; per block the LAST wave counts")
for e, n in names.items():
    v = rel[:, :, e].max(1)           # the slowest wave of each block reaches the event
    m = v >= 0
    if m.any():
        print(f"{n:18s} ev {e:2d}: blocks={int(m.sum()):4d} mean {v[m].mean():9.0f}  p50 {np.median(v[m]):9.0f}  max {v[m].max():8d}")
ends = rel[:, :, 23].max(1)
print("block duration: min", ends.min(), "mean", int(ends.mean()), "p90", int(np.percentile(ends, 90)), "max", ends.max())
worst = int(np.argmax(ends))
print("slowest block", worst, {e: int(rel[worst, :, e].max()) for e in names})
print("blocks with a second round:", int((rel[:, :, 9] >= 0).any(1).sum()))
# per WAVE (not per block): how long one wave spends between two of its own stamps
for a, b, n in ((2, 3, "staging (list barrier -> rows staged)"), (3, 4, "phase A (rows staged -> phase A done)"),
                (2, 4, "list barrier -> phase A done"), (5, 6, "phase B (Tp barrier -> phase B done)")):
    m = (rel[:, :, a] >= 0) & (rel[:, :, b] >= 0)
    d = (rel[:, :, b] - rel[:, :, a])[m]
    if d.size:
        print(f"  per wave {n:40s}: waves {d.size:5d} mean {d.mean():8.0f}  p50 {np.median(d):8.0f}  p90 {np.percentile(d, 90):8.0f}  max {d.max():8d}")
# per wave of the slowest blocks: when each wave finished phase A, phase B (round 0) and reached the final barrier
order = np.argsort(-ends)[:4]
for b in list(order) + [int(np.argsort(ends)[len(ends) // 2])]:
    print(f"block {b}: duration {ends[b]}; per wave  staged {rel[b, :, 3].tolist()}  phase A done {rel[b, :, 4].tolist()}  phase B done {rel[b, :, 6].tolist()}  "
          f"round end {rel[b, :, 7].tolist()}  before final {rel[b, :, 21].tolist()}  summed {rel[b, :, 22].tolist()}")
