"""Dump the reference kernels' GeometryState and this library's geom arrays for one scene (diagnostic)."""
import os, sys, types
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "scripts", "diag"))
import numpy as np
import util
from oracle import ref_cuda
import importlib.util
spec = importlib.util.spec_from_file_location("pb", os.path.join(ROOT, "scripts", "diag", "preprocess_bits_lib.py"))
pb = importlib.util.module_from_spec(spec); spec.loader.exec_module(pb)
NEG = (sys.argv[2] != "pos") if len(sys.argv) > 2 else True
case = dict(P=100000, F=32, W=128, H=128, neg=NEG, bg=(0.0, 0.0, 0.0), seed=0, cam_index=0)
sc, cam, kw, dC, dF = util.scene_case(**case)
from manigaussian_amd import synthetic as syn
PHASE = float(sys.argv[1]) if len(sys.argv) > 1 else 0.0
if PHASE:  # a generic camera: the test cameras sit at multiples of 90 degrees, their view matrices hold exact zeros
    cam = syn.circle_cameras(4, case["W"], case["H"], negative_focal=case["neg"], phase=PHASE)[1]
    if len(sys.argv) > 3:  # a ROLLED camera (tilted up vector): no entry of the view matrix is (nearly) zero
        import math
        import numpy as np
        th = PHASE + math.pi / 2
        target = np.array([0.2, 0.0, 0.9]); eye = target + np.array([1.3 * math.cos(th), 1.3 * math.sin(th), 0.9])
        c2w = syn.look_at_c2w(eye, target, up=(0.35, 0.2, 1.0), flip_xy=case["neg"])
        f = (case["W"] / 2) / math.tan(math.radians(20.0)); fs = -f if case["neg"] else f
        K = np.array([[fs, 0, case["W"] / 2], [0, fs, case["H"] / 2], [0, 0, 1]], np.float64)
        cam = syn.novel_calib(c2w, K, case["W"], case["H"])
    kw = syn.camera_settings_kwargs(cam, 1, True, bg=case["bg"])
st = types.SimpleNamespace(**kw)
ref = ref_cuda.forward_geometry(sc["means3D"], sc["opacities"], st, shs=sc["shs"], language_feature=sc["language_feature"],
                                scales=sc["scales"], rotations=sc["rotations"])
hip = pb.hip_geometry(sc, cam, case)
np.savez_compressed(os.path.join(ROOT, "gpurun_out", "ref_geom%s%s%s.npz" % ("_generic" if PHASE else "", "" if NEG else "_pos", "_rolled" if len(sys.argv) > 3 else "")), scales=sc["scales"].numpy(), rot=sc["rotations"].numpy(),
                    means3D=sc["means3D"].numpy(), opac=sc["opacities"].numpy(), vm=kw["viewmatrix"].numpy(), pm=kw["projmatrix"].numpy(),
                    tanfov=np.array([kw["tanfovx"], kw["tanfovy"]], np.float64),
                    **{"ref_" + k: v for k, v in ref.items() if k != "num_rendered"}, **{"hip_" + k: v for k, v in hip.items()})
print("saved")
