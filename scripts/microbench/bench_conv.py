"""Micro-benchmark of the conv kernel on representative Accel layer shapes (ReLU'd random data,
clocks warmed by a burst of launches before timing).

    python scripts/microbench/bench_conv.py 0,10,3,8        # tile ids, see launch_conv_igemm()

Tile ids 20-30 are the ablation builds of conv_igemm.hip (no global loads / no barrier / loads only /
LDS stores only) used for the breakdown quoted in DESIGN.md."""
import sys, numpy as np
sys.path.insert(0, __import__('os').path.join(__import__('os').path.dirname(__import__('os').path.abspath(__file__)), '..', '..'))
from accel_amd import runtime
if __import__('os').environ.get('ACCEL_LIB_PATH'):   # A/B two builds of the library on the same box
    runtime.LIB_PATH = __import__('os').environ['ACCEL_LIB_PATH']
SHAPES = [  # name, cin, cout, H, W (input), k, s, p, d, res, mode
    ("res4_2b 3x3 256-256 @64x128", 256, 256, 64, 128, 3, 1, 1, 1, 0, "conv"),
    ("res4_2a 1x1 1024-256", 1024, 256, 64, 128, 1, 1, 0, 1, 0, "conv"),
    ("res4_2c 1x1 256-1024 +res", 256, 1024, 64, 128, 1, 1, 0, 1, 1, "conv"),
    ("res3_2c 1x1 128-512 +res @128x256", 128, 512, 128, 256, 1, 1, 0, 1, 1, "conv"),
    ("res2_2c 1x1 64-256 +res @256x512", 64, 256, 256, 512, 1, 1, 0, 1, 1, "conv"),
    ("res2_2b 3x3 64-64 @256x512", 64, 64, 256, 512, 3, 1, 1, 1, 0, "conv"),
    ("fc6 1x1 2048-1024", 2048, 1024, 64, 128, 1, 1, 0, 1, 0, "conv"),
    ("res5_1 1x1 1024-2048", 1024, 2048, 64, 128, 1, 1, 0, 1, 0, "conv"),
    ("conv1 7x7 3-64 s2 @1024x2048", 3, 64, 1024, 2048, 7, 2, 3, 1, 0, "conv"),
    ("flow conv2 5x5 64-128 s2 @256x512", 64, 128, 256, 512, 5, 2, 2, 1, 0, "conv"),
    ("feat_up deconv 512-2048 @32x64", 512, 2048, 32, 64, 4, 2, 1, 1, 0, "deconv2x"),
    ("r18 3x3 128-128 @128x256", 128, 128, 128, 256, 3, 1, 1, 1, 0, "conv"),
]
if __import__('os').environ.get("STEADY"):   # long-K shapes: steady-state main loop with 2 / 4 / 8 blocks of 64x64 per CU
    SHAPES = [
        ("K=8192 N=256 M=8192 (2 blk/CU @64x64)", 8192, 256, 64, 128, 1, 1, 0, 1, 0, "conv"),
        ("K=8192 N=256 M=16384 (4 blk/CU)", 8192, 256, 128, 128, 1, 1, 0, 1, 0, "conv"),
        ("K=8192 N=512 M=16384 (8 blk/CU)", 8192, 512, 128, 128, 1, 1, 0, 1, 0, "conv"),
        ("K=2304 N=256 M=8192  1x1", 2304, 256, 64, 128, 1, 1, 0, 1, 0, "conv"),
        ("K=2304 N=256 M=8192  3x3", 256, 256, 64, 128, 3, 1, 1, 1, 0, "conv"),
        ("K=2304 N=256 M=32768 3x3", 256, 256, 128, 256, 3, 1, 1, 1, 0, "conv"),
    ]
if __import__('os').environ.get("WINO"):     # the 3x3 / stride-1 layers of the Accel-18 step, at 1 and at 8 clips per call (rows stacked)
    base = [("res2_2b 3x3 64-64 @256x512", 64, 64, 256, 512, 0), ("r18 s1 3x3 64-64 +res", 64, 64, 256, 512, 1),
            ("res3_2b 3x3 128-128 @128x256", 128, 128, 128, 256, 0), ("res4_2b 3x3 256-256 @64x128", 256, 256, 64, 128, 0),
            ("r18 res5 3x3 512-512 @32x64", 512, 512, 32, 64, 0), ("flow conv5_1 3x3 512-512 @16x32", 512, 512, 16, 32, 0),
            ("flow conv6_1 3x3 1024-1024 @8x16", 1024, 1024, 8, 16, 0)]
    SHAPES = []
    for (n, ci, co, h, w, rs) in base:
        SHAPES.append((n + " x1", ci, co, h, w, 3, 1, 1, 1, rs, "conv"))
        SHAPES.append((n + " x8", ci, co, 8 * h, w, 3, 1, 1, 1, rs, "conv"))
if __import__('os').environ.get("SHORTK"):   # short-K residual 1x1 layers at 8 clips per call (rows stacked)
    SHAPES = [("res4_2c 1x1 256-1024 +res x8", 256, 1024, 512, 128, 1, 1, 0, 1, 1, "conv"),
              ("res3_2c 1x1 128-512 +res x8", 128, 512, 1024, 256, 1, 1, 0, 1, 1, "conv"),
              ("res2_2c 1x1 64-256 +res x8", 64, 256, 2048, 512, 1, 1, 0, 1, 1, "conv"),
              ("res4_2a 1x1 1024-256 x8", 1024, 256, 512, 128, 1, 1, 0, 1, 0, "conv")]
if __import__('os').environ.get("B8"):       # the layers that carry the 8-clips-per-call step (rows of the 8 images stacked)
    SHAPES = [("fc6 1x1 2048-1024 x8", 2048, 1024, 512, 128, 1, 1, 0, 1, 0, "conv"),
              ("res5a_1 1x1 1024-2048 x8", 1024, 2048, 512, 128, 1, 1, 0, 1, 0, "conv"),
              ("res5 2b cols 4608-512 x8", 4608, 512, 512, 128, 1, 1, 0, 1, 0, "conv"),
              ("res5 2c 1x1 512-2048 +res x8", 512, 2048, 512, 128, 1, 1, 0, 1, 1, "conv"),
              ("res5 2a 1x1 2048-512 x8", 2048, 512, 512, 128, 1, 1, 0, 1, 0, "conv"),
              ("res4_2a 1x1 1024-256 x8", 1024, 256, 512, 128, 1, 1, 0, 1, 0, "conv"),
              ("res4_2c 1x1 256-1024 +res x8", 256, 1024, 512, 128, 1, 1, 0, 1, 1, "conv"),
              ("res4_2b 3x3 256-256 x8", 256, 256, 512, 128, 3, 1, 1, 1, 0, "conv"),
              ("res3_2a 1x1 512-128 x8", 512, 128, 1024, 256, 1, 1, 0, 1, 0, "conv"),
              ("res3_2b 3x3 128-128 x8", 128, 128, 1024, 256, 3, 1, 1, 1, 0, "conv"),
              ("res3_2c 1x1 128-512 +res x8", 128, 512, 1024, 256, 1, 1, 0, 1, 1, "conv"),
              ("res2_2a 1x1 256-64 x8", 256, 64, 2048, 512, 1, 1, 0, 1, 0, "conv"),
              ("feat_up*fc6 deconv 512-1024 @32x64 x8", 512, 1024, 256, 64, 4, 2, 1, 1, 0, "deconv2x"),
              ("flow conv2 5x5 64-128 s2 x8", 64, 128, 2048, 512, 5, 2, 2, 1, 0, "conv"),
              ("flow conv3 5x5 128-256 s2 x8", 128, 256, 1024, 256, 5, 2, 2, 1, 0, "conv"),
              ("r18 s2 3x3 64-128 s2 x8", 64, 128, 2048, 512, 3, 2, 1, 1, 0, "conv"),
              ("r18 res5a 2b cols 4608-512 @32x64 x8", 4608, 512, 256, 64, 1, 1, 0, 1, 0, "conv")]
if __import__('os').environ.get("KSWEEP"):   # time vs K at fixed M, N: slope = steady-state rate, intercept = fixed cost per launch
    SHAPES = [("K=%d N=256 M=8192" % k, k, 256, 64, 128, 1, 1, 0, 1, 0, "conv") for k in (32, 64, 128, 256, 512, 1024, 2304, 4608, 8192)]
    SHAPES += [("K=%d N=1024 M=8192" % k, k, 1024, 64, 128, 1, 1, 0, 1, 0, "conv") for k in (32, 256, 1024, 4096)]
if __import__('os').environ.get("BATCH4"):   # what would batching 4 frames buy?  each shape at 1x and at 4x the rows
    base = [("res2_2b 3x3 64-64 @256x512", 64, 64, 256, 512, 3, 1, 1, 1, 0), ("r18 s1 3x3 64-64 +res", 64, 64, 256, 512, 3, 1, 1, 1, 1),
            ("res2_2a 1x1 256-64", 256, 64, 256, 512, 1, 1, 0, 1, 0),
            ("res4_2a 1x1 1024-256", 1024, 256, 64, 128, 1, 1, 0, 1, 0), ("res4_2b 3x3 256-256", 256, 256, 64, 128, 3, 1, 1, 1, 0),
            ("res4_2c 1x1 256-1024+res", 256, 1024, 64, 128, 1, 1, 0, 1, 1), ("res3_2b 3x3 128-128", 128, 128, 128, 256, 3, 1, 1, 1, 0),
            ("res3_2c 1x1 128-512+res", 128, 512, 128, 256, 1, 1, 0, 1, 1), ("res5_2a 1x1 2048-512", 2048, 512, 64, 128, 1, 1, 0, 1, 0),
            ("r18 s3 3x3 256-256", 256, 256, 64, 128, 3, 1, 1, 1, 1), ("r18 res5 3x3 512-512 @32x64", 512, 512, 32, 64, 3, 1, 1, 1, 0),
            ("flow conv4_1 3x3 512-512 @32x64", 512, 512, 32, 64, 3, 1, 1, 1, 0), ("flow conv5_1 3x3 512-512 @16x32", 512, 512, 16, 32, 3, 1, 1, 1, 0),
            ("flow conv6_1 3x3 1024-1024 @8x16", 1024, 1024, 8, 16, 3, 1, 1, 1, 0)]
    SHAPES = []
    for (n, ci, co, h, w, k, st, pd, dl, rs) in base:
        SHAPES.append((n + " x1", ci, co, h, w, k, st, pd, dl, rs, "conv"))
        SHAPES.append((n + " x4", ci, co, 4 * h, w, k, st, pd, dl, rs, "conv"))
if __import__('os').environ.get("STRIP"):    # layers with at most 72 output channels at 8 clips per call: the offset branches of res5, the score layer
    SHAPES = [("res5 offset 3x3 d2 512-18 x8", 512, 18, 512, 128, 3, 1, 2, 2, 0, "conv"),
              ("r18 res5 offset 3x3 d2 512-72 @32x64 x8", 512, 72, 256, 64, 3, 1, 2, 2, 0, "conv"),
              ("res5 offset 3x3 d2 512-18 x1", 512, 18, 64, 128, 3, 1, 2, 2, 0, "conv"),
              ("r18 res5 offset 3x3 d2 512-72 @32x64 x1", 512, 72, 32, 64, 3, 1, 2, 2, 0, "conv"),
              ("res5 offset 3x3 d1 512-18 x8", 512, 18, 512, 128, 3, 1, 1, 1, 0, "conv"),
              ("score 1x1 1024-19 x8", 1024, 19, 512, 128, 1, 1, 0, 1, 0, "conv")]
if __import__('os').environ.get("ONLY"):     # keep the shapes whose name holds one of the comma-separated fragments
    SHAPES = [sh for sh in SHAPES if any(f in sh[0] for f in __import__('os').environ["ONLY"].split(","))]
ctx = runtime.Context(0)
tiles = [int(t) for t in sys.argv[1].split(',')] if len(sys.argv) > 1 else [-1]
for (name, cin, cout, H, W, k, s, p, d, res, mode) in SHAPES:
    if mode == "deconv2x":
        Ho, Wo = 2 * H, 2 * W
    else:
        Ho = (H + 2 * p - d * (k - 1) - 1) // s + 1; Wo = (W + 2 * p - d * (k - 1) - 1) // s + 1
    cp, kp = (cin + 3) // 4 * 4, (cout + 3) // 4 * 4
    al = lambda b: (b + 255) // 256 * 256
    o_x = 0; o_y = al(H * W * cp * 4); o_r = o_y + al(Ho * Wo * kp * 4); tot = o_r + al(Ho * Wo * kp * 4)
    flops = 2.0 * Ho * Wo * cout * cin * k * k if mode == "conv" else 2.0 * H * W * cout * cin * 16
    line = []
    for tile in tiles:
        m = runtime.Model(ctx)
        rng = np.random.default_rng(0)
        wshape = (cout, cin, k, k) if mode == "conv" else (cin, cout, 4, 4)
        m.set_param("w_weight", (rng.standard_normal(wshape) * 0.05).astype(np.float32))
        t = "option graph=0\narena bytes=%d\npbuf name=x bytes=%d\npbuf name=r bytes=%d\n" % (tot, cin * H * W * 4, cout * Ho * Wo * 4)
        t += "import_nchw src=x:0:%d:%d:%d:%d dst=A:%d:%d:%d:%d:%d\n" % (cin, cin, H, W, o_x, cin, cp, H, W)
        t += "import_nchw src=r:0:%d:%d:%d:%d dst=A:%d:%d:%d:%d:%d\n" % (cout, cout, Ho, Wo, o_r, cout, kp, Ho, Wo)
        t += "conv name=c in=A:%d:%d:%d:%d:%d out=A:%d:%d:%d:%d:%d w=w_weight act=1 cin=%d cout=%d mode=%s tile=%d flops=%g" % (
            o_x, cin, cp, H, W, o_y, cout, kp, Ho, Wo, cin, cout, mode, tile, flops)
        if mode == "conv":
            t += " k=%d,%d s=%d,%d p=%d,%d d=%d,%d" % (k, k, s, s, p, p, d, d)
        if res:
            t += " res=A:%d:%d:%d:%d:%d" % (o_r, cout, kp, Ho, Wo)
        t += "\n"
        plan = m.add_plan("b", t)
        xin = np.maximum(rng.standard_normal((cin, H, W)), 0).astype(np.float32)
        if __import__('os').environ.get("XDATA") == "zero":      # power / clock experiment: the same launch on all-zero pixels
            xin[:] = 0
        elif __import__('os').environ.get("XDATA") == "dense":   # no ReLU zeros: every operand bit toggles
            xin = rng.standard_normal((cin, H, W)).astype(np.float32)
        m.write("x", xin)
        m.write("r", rng.standard_normal((cout, Ho, Wo)).astype(np.float32))
        plan.finalize()
        for _ in range(60): plan.run()      # warm the clocks: short isolated launches under-clock
        ctx.sync()
        ms = plan.profile(20)[2]
        line.append("t%d %7.1f us %6.1f TF" % (tile, ms * 1e3, flops / ms / 1e9))
        m.close()
    print("%-36s %s" % (name, " | ".join(line)), flush=True)
