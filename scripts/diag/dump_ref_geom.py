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
case = dict(P=100000, F=32, W=128, H=128, neg=True, bg=(0.0, 0.0, 0.0), seed=0, cam_index=0)
sc, cam, kw, dC, dF = util.scene_case(**case)
from manigaussian_amd import synthetic as syn
PHASE = float(sys.argv[1]) if len(sys.argv) > 1 else 0.0
if PHASE:  # a generic camera: the test cameras sit at multiples of 90 degrees, their view matrices hold exact zeros
    cam = syn.circle_cameras(4, case["W"], case["H"], negative_focal=case["neg"], phase=PHASE)[1]
    kw = syn.camera_settings_kwargs(cam, 1, True, bg=case["bg"])
st = types.SimpleNamespace(**kw)
ref = ref_cuda.forward_geometry(sc["means3D"], sc["opacities"], st, shs=sc["shs"], language_feature=sc["language_feature"],
                                scales=sc["scales"], rotations=sc["rotations"])
hip = pb.hip_geometry(sc, cam, case)
np.savez_compressed(os.path.join(ROOT, "gpurun_out", "ref_geom%s.npz" % ("_generic" if PHASE else "")), scales=sc["scales"].numpy(), rot=sc["rotations"].numpy(),
                    means3D=sc["means3D"].numpy(), opac=sc["opacities"].numpy(), vm=kw["viewmatrix"].numpy(), pm=kw["projmatrix"].numpy(),
                    tanfov=np.array([kw["tanfovx"], kw["tanfovy"]], np.float64),
                    **{"ref_" + k: v for k, v in ref.items() if k != "num_rendered"}, **{"hip_" + k: v for k, v in hip.items()})
print("saved")
