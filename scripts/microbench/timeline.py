"""Where does a conv launch spend its time?  Needs the diagnostic build of the library:

    make -C accel_amd/csrc timeline        # -> scratch/libaccel_tl.so  (-DACCEL_CONV_TIMELINE)
    python scripts/microbench/timeline.py <tile> cin cout H W k

Per block: entry, first tile in LDS, K loop done, stores retired (100 MHz constant clock, 10 ns ticks).
Printed relative to the earliest block entry of the launch."""
import ctypes, os, sys
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, '..', '..'))
from accel_amd import runtime
runtime.LIB_PATH = os.path.join(HERE, '..', '..', 'scratch', 'libaccel_tl.so')
tile = int(sys.argv[1]) if len(sys.argv) > 1 else 8
cin, cout, H, W, k = [int(v) for v in sys.argv[2:7]] if len(sys.argv) > 6 else (256, 256, 64, 128, 3)
pad = k // 2
ctx = runtime.Context(0)
m = runtime.Model(ctx)
rng = np.random.default_rng(0)
m.set_param("w_weight", (rng.standard_normal((cout, cin, k, k)) * 0.05).astype(np.float32))
al = lambda b: (b + 255) // 256 * 256
o_y = al(H * W * cin * 4)
t = "option graph=0\narena bytes=%d\npbuf name=x bytes=%d\n" % (o_y + al(H * W * cout * 4), cin * H * W * 4)
t += "import_nchw src=x:0:%d:%d:%d:%d dst=A:0:%d:%d:%d:%d\n" % (cin, cin, H, W, cin, cin, H, W)
t += "conv name=c in=A:0:%d:%d:%d:%d out=A:%d:%d:%d:%d:%d w=w_weight act=1 cin=%d cout=%d mode=conv tile=%d k=%d,%d s=1,1 p=%d,%d d=1,1\n" % (
    cin, cin, H, W, o_y, cout, cout, H, W, cin, cout, tile, k, k, pad, pad)
plan = m.add_plan("b", t)
m.write("x", np.maximum(rng.standard_normal((cin, H, W)), 0).astype(np.float32))
plan.finalize()
for _ in range(50):
    plan.run()
ctx.sync()
nblk = int(os.environ.get("NBLK", "512"))
buf = np.zeros(8 * 16384, np.uint64)
L = runtime.lib()
L.accel_debug_conv_timeline.argtypes = [ctypes.c_void_p, ctypes.c_int]
for rep in range(3):
    plan.run(); ctx.sync()
    rc = L.accel_debug_conv_timeline(buf.ctypes.data_as(ctypes.c_void_p), 8 * 16384)
    assert rc == 0
    tl = buf[:8 * nblk].reshape(nblk, 8).astype(np.int64)
    tl = (tl - tl[:, 0].min()) * 0.01        # us
    q = lambda a: "min %6.2f  p10 %6.2f  med %6.2f  p90 %6.2f  max %6.2f" % (a.min(), np.percentile(a, 10), np.median(a), np.percentile(a, 90), a.max())
    print("rep %d  blocks %d" % (rep, nblk))
    print("  entry (launch skew)        ", q(tl[:, 0]))
    print("  prologue (entry->tile 0)   ", q(tl[:, 1] - tl[:, 0]))
    print("  K loop                     ", q(tl[:, 2] - tl[:, 1]))
    print("  epilogue (stores retired)  ", q(tl[:, 3] - tl[:, 2]))
    print("  block end                  ", q(tl[:, 3]))
    nk = (cin * k * k + 31) // 32
    marks = [(0, tl[:, 1]), (nk // 8, tl[:, 4]), (nk // 4, tl[:, 5]), (nk // 2, tl[:, 6]), (3 * nk // 4, tl[:, 7]), (nk, tl[:, 2])]
    for (k0, t0), (k1, t1) in zip(marks, marks[1:]):
        if k1 > k0:
            print("  K steps %4d..%4d: %.3f us per step (median over blocks)" % (k0, k1, float(np.median(t1 - t0)) / (k1 - k0)))
