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
import util  # noqa: E402
from manigaussian_amd import GaussianRasterizationSettings, GaussianRasterizer  # noqa: E402
from manigaussian_amd import synthetic as syn  # noqa: E402

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
mg.check_status(dev)
assert torch.equal(out[0], eager[0]) and torch.equal(out[1], eager[1]) and torch.equal(out[2], eager[2]), "images differ"
for a, b in zip(out[3:], eager[3:]):
    assert (a - b).abs().max().item() <= 2e-5 * b.abs().max().item() + 1e-12, "gradients differ"
# the graph reads the parameters where they live: an in-place update is seen by the next replay
with torch.no_grad():
    leaves["means3D"].add_(0.01)
graph.replay()
torch.cuda.synchronize()
replayed = [t.clone() for t in out[:3]]
moved = [t.detach() for t in step()]
torch.cuda.synchronize()
assert torch.equal(replayed[0], moved[0]) and torch.equal(replayed[2], moved[2]), "replay does not follow the parameters"
assert not torch.equal(moved[0], eager[0])
mg.check_status(dev)
print("GRAPH_OK")
