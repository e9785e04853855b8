"""Phase timeline of the bucket-rank binning kernel (diagnostic): python scripts/trace_bin.py [P] [W]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from manigaussian_amd import GaussianRasterizationSettings, GaussianRasterizer, _lib
from manigaussian_amd import synthetic as syn

dev = torch.device("cuda:0")
P = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
W = int(sys.argv[2]) if len(sys.argv) > 2 else 128
F = 32
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
EV = 16
buf = np.zeros(1024 * EV, np.uint64)
rc = L.mgs_debug_read_trace_bin(buf.ctypes.data, buf.size)
assert rc == 0, rc
t = buf.reshape(1024, EV).astype(np.int64)
live = t[:, 15] > 0
t = t[live]
rel = np.where(t > 0, t - t[:, :1], -1)
names = {0: "entry", 1: "range read", 2: "keys in registers, depth min/max", 3: "bounds reduced", 4: "level-1 histogram",
         5: "level-1 scan + rounds", 6: "round's keys marked", 8: "level-2 histogram", 9: "level-2 scan", 10: "keys grouped",
         11: "owners ranked", 12: "ids stored", 15: "exit"}
print(f"P={P} W={W}: {int(live.sum())} live workgroups; shader-clock cycles (s_memtime) since the workgroup's entry (wave 0)")
for e, n in names.items():
    v = rel[:, e]
    m = v >= 0
    if m.any():
        print(f"{n:28s} ev {e:2d}: wgs={int(m.sum()):4d} mean {v[m].mean():9.0f}  p50 {np.median(v[m]):9.0f}  max {v[m].max():8d}")
worst = int(np.argmax(rel[:, 15]))
print("slowest workgroup", {e: int(rel[worst, e]) for e in names})
