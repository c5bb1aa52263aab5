import os, sys, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
os.environ["ACCEL_CONV_DTYPE"] = "f16"
from accel_amd import runtime
from oracle import ops as O
ctx = runtime.Context(0)
h = lambda a: a.astype(np.float16).astype(np.float32)
rng = np.random.default_rng(0)
for (C, K, H, W, k, s, p, d) in [(64, 136, 23, 31, 3, 1, 1, 1), (256, 72, 9, 13, 1, 1, 0, 1), (512, 256, 16, 24, 3, 1, 1, 1)]:
    x = rng.standard_normal((1, C, H, W)).astype(np.float32)
    w = (rng.standard_normal((K, C, k, k)) * (2.0 / (C * k * k)) ** 0.5).astype(np.float32)
    for tile in (-1, 0, 3, 10):
        got = ctx.conv2d(x, w, None, s, p, d, tile=tile)
        r16 = O.conv2d(h(x), h(w), None, s, p, d)
        r32 = O.conv2d(x, w, None, s, p, d)
        sc = float(np.abs(r32).max())
        print("C %d K %d k %d tile %3d: |got - rounded-operand oracle| %.2e   |got - fp32 oracle| %.2e   (of the output range)" % (
            C, K, k, tile, float(np.abs(got - r16).max()) / sc, float(np.abs(got - r32).max()) / sc))
