"""Randomised sweep of the view-batch entry point against V single-view calls (test infrastructure):
python tests/tools/fuzz_views.py [N] [seed].  Every case is tests/test_gpu_parity.py's _view_batch_equals_per_view_calls (images
and radii bit for bit, gradients summed over the views) at a random Gaussian count, view count, image size and feature width."""
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_parity as tp  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    bad = 0
    for i in range(n):
        case = dict(P=int(10 ** rng.uniform(0.0, 4.8)), F=rng.choice([3, 3, 8, 32]), V=rng.choice([2, 3, 4, 5, 8]),
                    precomp=rng.random() < 0.2)
        case["W"], case["H"] = rng.choice([(8, 8), (17, 33), (32, 32), (40, 72), (64, 64), (100, 52), (128, 128), (200, 120)])
        try:
            tp._view_batch_equals_per_view_calls(case)
            print(f"OK  #{i} {case}", flush=True)
        except AssertionError as e:
            bad += 1
            print(f"BAD #{i} {case}: {str(e)[:300]}", flush=True)
    print(f"{n - bad}/{n} cases: the batch equals the per-view calls")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
