"""Runs one conv shape many times (for PC sampling / counters): python scripts/microbench/one_conv.py <tile> [cin cout H W k]"""
import sys, os, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
from accel_amd import runtime
tile = int(sys.argv[1]) if len(sys.argv) > 1 else 8
cin, cout, H, W, k = [int(v) for v in sys.argv[2:7]] if len(sys.argv) > 6 else (256, 256, 64, 128, 3)
p = k // 2
ctx = runtime.Context(0)
m = runtime.Model(ctx)
rng = np.random.default_rng(0)
m.set_param("w_weight", (rng.standard_normal((cout, cin, k, k)) * 0.05).astype(np.float32))
al = lambda b: (b + 255) // 256 * 256
o_y = al(H * W * cin * 4)
t = "option graph=0\narena bytes=%d\npbuf name=x bytes=%d\n" % (o_y + al(H * W * cout * 4), cin * H * W * 4)
t += "import_nchw src=x:0:%d:%d:%d:%d dst=A:0:%d:%d:%d:%d\n" % (cin, cin, H, W, cin, cin, H, W)
t += "conv name=c in=A:0:%d:%d:%d:%d out=A:%d:%d:%d:%d:%d w=w_weight act=1 cin=%d cout=%d mode=conv tile=%d k=%d,%d s=1,1 p=%d,%d d=1,1\n" % (
    cin, cin, H, W, o_y, cout, cout, H, W, cin, cout, tile, k, k, p, p)
plan = m.add_plan("b", t)
m.write("x", np.maximum(rng.standard_normal((cin, H, W)), 0).astype(np.float32))
plan.finalize()
for _ in range(int(os.environ.get("ITERS", "300"))):
    plan.run()
ctx.sync()
print("done")
