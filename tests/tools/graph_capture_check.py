"""Run in its OWN process by tests/test_gpu_parity.py::test_forward_backward_captured_into_a_hip_graph_replays_bit_identically
(stream capture is process-wide state; a capture that goes wrong takes the process with it, not the test session).

fwd + bwd through the public autograd API captured with torch.cuda.graph: the library's forward is asynchronous (no host
synchronisation inside), so the capture succeeds; replays reproduce the eager images bit for bit, follow in-place parameter
updates, and report through check_status().  Prints GRAPH_OK on success."""
import faulthandler
import os
import sys

faulthandler.enable()
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

import manigaussian_amd as mg  # noqa: E402

mg.set_forward_mode("async")  # graph capture needs forwards that never synchronise (opt-in; the default is "safe")
import util  # noqa: E402
from manigaussian_amd import GaussianRasterizationSettings, GaussianRasterizer  # noqa: E402
from manigaussian_amd import synthetic as syn  # noqa: E402

if os.environ.get("MGS_GM_WAVES"):
    from manigaussian_amd import _lib as _l
    _l.set_option("gm_waves", int(os.environ["MGS_GM_WAVES"]))
dev = torch.device("cuda:0")
P, F, W = int(sys.argv[1]) if len(sys.argv) > 1 else 20000, 32, 128
sc, cam, kw, dC, dF = util.scene_case(P=P, F=F)
leaves = {k: v.to(dev).requires_grad_(True) for k, v in sc.items()}
rast = GaussianRasterizer(GaussianRasterizationSettings(**syn.camera_settings_kwargs(cam, 1, True, device=dev)))
dC, dF = dC.to(dev), dF.to(dev)
m2 = torch.zeros(P, 3, device=dev)


def step():
    c, f, r = rast(leaves["means3D"], m2, leaves["opacities"], shs=leaves["shs"],
                   language_feature_precomp=leaves["language_feature"], scales=leaves["scales"],
                   rotations=leaves["rotations"])
    return (c, f, r) + torch.autograd.grad([c, f], list(leaves.values()), [dC, dF])


def stage(msg):
    print("stage:", msg, flush=True)


for _ in range(3):
    # detached copies: a kept output would keep its autograd graph -- and the leaves' AccumulateGrad nodes, bound to the
    # default stream -- alive into the capture, where torch would then try to synchronise the two streams (its own
    # warning says so) and the runtime aborts the capture
    eager = [t.detach().clone() for t in step()]
    mg.check_status(dev)
stage("eager steps done")
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    step()                                   # warm-up on a side stream, as torch's capture recipe asks
torch.cuda.current_stream().wait_stream(side)
torch.cuda.synchronize()
stage("side-stream warm-up done")
graph = torch.cuda.CUDAGraph()
with torch.cuda.graph(graph):
    out = step()
stage("captured")
for _ in range(3):
    graph.replay()
torch.cuda.synchronize()
stage("replayed")
SKIP = os.environ.get("MGS_GRAPH_SKIP", "")
if "status" not in SKIP:
    mg.check_status(dev)
stage("status checked")
if "images" not in SKIP:
    assert torch.equal(out[0], eager[0]) and torch.equal(out[1], eager[1]) and torch.equal(out[2], eager[2]), "images differ"
if "grads" not in SKIP:
    for a, b in zip(out[3:], eager[3:]):
        assert (a - b).abs().max().item() <= 2e-5 * b.abs().max().item() + 1e-12, "gradients differ"
# the graph reads the parameters where they live: an in-place update is seen by the next replay
stage("replays equal the eager step")
if os.environ.get("MGS_GRAPH_DUMP"):
    from manigaussian_amd import _state
    st_ = _state.device_state(dev)
    snap = torch.cuda.memory_snapshot()

    def where(ptr):
        for seg in snap:
            if seg["address"] <= ptr < seg["address"] + seg["total_size"]:
                off = seg["address"]
                for b in seg["blocks"]:
                    if off <= ptr < off + b["size"]:
                        return f"pool {seg.get('segment_pool_id')} {seg.get('segment_type')} block {b['state']} size {b['size']}"
                    off += b["size"]
        return "NOT IN ANY DEVICE SEGMENT"
    a_ = st_.captured[0].a
    for name in ("background", "means3D", "shs", "language_feature", "opacities", "scales", "rotations", "viewmatrix",
                 "projmatrix", "campos", "geom", "binning", "img", "bwd_accum"):
        v = getattr(a_, name)
        print("arg", name, hex(v or 0), where(v or 0), flush=True)
    for i, t in enumerate(out):
        print("out", i, hex(t.data_ptr()), where(t.data_ptr()), flush=True)
    print("dC", where(dC.data_ptr()), "dF", where(dF.data_ptr()), flush=True)
MOVE = float(os.environ.get("MGS_GRAPH_MOVE", "0.01"))
if os.environ.get("MGS_GRAPH_EAGER_FIRST"):
    with torch.no_grad():
        leaves["means3D"].add_(MOVE)
    step()
    torch.cuda.synchronize()
    stage("eager step on moved parameters before the replay: fine")
    with torch.no_grad():
        leaves["means3D"].sub_(MOVE)
with torch.no_grad():
    leaves["means3D"].add_(MOVE)
torch.cuda.synchronize()
stage("parameters moved")
graph.replay()
torch.cuda.synchronize()
stage("replayed on the moved parameters")
replayed = [t.clone() for t in out[:3]]
moved = [t.detach() for t in step()]
torch.cuda.synchronize()
stage("eager step on the moved parameters")
assert torch.equal(replayed[0], moved[0]) and torch.equal(replayed[2], moved[2]), "replay does not follow the parameters"
assert not torch.equal(moved[0], eager[0])
mg.check_status(dev)
print("GRAPH_OK")
