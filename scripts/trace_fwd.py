"""Phase timeline of the dense forward at C3 (diagnostic): python scripts/trace_fwd.py [dense_variant]"""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from manigaussian_amd import GaussianRasterizationSettings, GaussianRasterizer, _lib
from manigaussian_amd import synthetic as syn

variant = int(sys.argv[1]) if len(sys.argv) > 1 else 0
dev = torch.device("cuda:0")
P, F, W = 100000, 32, 128
sc = {k: v.to(dev) for k, v in syn.make_scene(P, F=F, M=4, seed=0).items()}
cam = syn.circle_cameras(8, W, W, negative_focal=True)[0]
rast = GaussianRasterizer(GaussianRasterizationSettings(**syn.camera_settings_kwargs(cam, 1, True, device=dev)))
_lib.set_option("fwd_mode", 2)
_lib.set_option("dense_variant", variant)


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
t0 = t[:, :, 0].min()
rel = np.where(t > 0, t - t0, -1)
names = {0: "entry", 1: "fill done r0", 2: "list barrier r0", 3: "rows staged r0", 4: "phase A done r0", 5: "Tp barrier r0",
         6: "phase B done r0", 7: "round end r0", 9: "fill done r1", 10: "list barrier r1", 11: "rows r1", 12: "phase A r1",
         13: "Tp barrier r1", 14: "phase B r1", 15: "round end r1", 21: "before final", 22: "image summed", 23: "exit"}
print(f"variant {variant}; ticks of s_memtime (100 MHz => 10 ns per tick if constant-rate)")
for e, n in names.items():
    v = rel[:, :, e]
    m = v >= 0
    if m.any():
        print(f"{n:18s} ev {e:2d}: n={int(m.sum()):5d} mean {v[m].mean():8.1f}  p50 {np.median(v[m]):8.1f}  max {v[m].max():8d}")
ends = rel[:, :, 23].max(1)
print("block end ticks: min", ends.min(), "mean", ends.mean(), "p90", np.percentile(ends, 90), "max", ends.max())
starts = rel[:, :, 0].min(1)
print("block start ticks: max", starts.max())
worst = int(np.argmax(ends))
print("worst block", worst, "per-event max over waves:", {e: int(rel[worst, :, e].max()) for e in names})
r2 = (rel[:, :, 9] >= 0).any(1).sum()
print("blocks with a second round:", int(r2))
