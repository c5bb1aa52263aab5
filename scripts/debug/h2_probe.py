"""Accuracy of the fp16x2 form (ACCEL_SPLIT=h2) of the bf16x3 kernels against float64, beside the bf16x3 form and plain fp32
accumulation (numpy): 1x1 layers of growing K on ReLU'd activations of different scales.
    python scripts/debug/h2_probe.py [tile]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
from accel_amd import runtime

tile = int(sys.argv[1]) if len(sys.argv) > 1 else 81
rng = np.random.default_rng(0)
for (C, K, H, W, xscale, wscale) in ((256, 128, 32, 64, 1.0, 0.05), (2048, 256, 32, 64, 1.0, 0.02), (2048, 256, 32, 64, 1e-3, 0.02), (2048, 256, 32, 64, 300.0, 1e-4),
                                     (4608, 128, 16, 32, 5.0, 0.01), (512, 128, 32, 64, 1e-5, 3.0)):
    x = (np.maximum(rng.standard_normal((1, C, H, W)), 0) * xscale * np.exp(rng.standard_normal((1, C, 1, 1)))).astype(np.float32)
    w = (rng.standard_normal((K, C, 1, 1)) * wscale).astype(np.float32)
    ref = np.einsum('kc,chw->khw', w[:, :, 0, 0].astype(np.float64), x[0].astype(np.float64))[None]
    f32 = np.einsum('kc,chw->khw', w[:, :, 0, 0], x[0])[None]
    out = {}
    for mode in ("b3", "h2"):
        os.environ["ACCEL_SPLIT"] = mode
        ctx = runtime.Context(0)
        out[mode] = ctx.conv2d(x, w, None, 1, 0, 1, tile=tile)
    s = np.abs(ref).max()
    e = lambda a: (np.abs(a - ref).max() / s, np.sqrt(np.mean((a - ref) ** 2)) / s)
    print("C=%4d K=%3d xs=%g ws=%g | numpy f32 max %.2e rms %.2e | bf16x3 max %.2e rms %.2e | fp16x2 max %.2e rms %.2e" % (
        (C, K, xscale, wscale) + e(f32) + e(out["b3"]) + e(out["h2"])), flush=True)
