"""Micro-reproducer: a convolution on the compute stream (launch geometry given: 74 = bf16x3 with LDS-DMA'd weights,
10 / 32 = fp32 MFMA, no LDS-DMA) beside a chain of exact copies (global_load_dwordx4 / global_store_dwordx4) on the
side stream.  Every copy must reproduce its source bit for bit.

    python scripts/debug/dma_vs_gload.py [tile] [runs] [copies] [eager|graph]
"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np

from accel_amd import runtime

tile = int(sys.argv[1]) if len(sys.argv) > 1 else 74
runs = int(sys.argv[2]) if len(sys.argv) > 2 else 20
K = int(sys.argv[3]) if len(sys.argv) > 3 else 40
graph = (sys.argv[4] if len(sys.argv) > 4 else "eager") == "graph"
C, Kc, H, W, NB = 2048, 1024, 64, 128, 4
CS = 512
rng = np.random.default_rng(0)
w = (rng.standard_normal((Kc, C, 1, 1)) * 0.03).astype(np.float32)
x = rng.standard_normal((NB, H, W, C)).astype(np.float32)
src = rng.standard_normal((H, W, CS)).astype(np.float32) + 3.0      # never zero
bad_runs = 0
for it in range(runs):
    ctx = runtime.Context(0)
    m = runtime.Model(ctx)
    m.set_param("a_weight", w)
    t = "%sarena bytes=256\n" % ("" if graph else "option graph=0\n")
    t += "pbuf name=x bytes=%d\npbuf name=y bytes=%d\npbuf name=s bytes=%d\n" % (x.nbytes, NB * H * W * Kc * 4, src.nbytes)
    for i in range(K):
        t += "pbuf name=d%d bytes=%d\n" % (i, src.nbytes)
    t += ("conv name=c0 in=x:0:%d:%d:%d:%d:%d out=y:0:%d:%d:%d:%d:%d w=a_weight act=1 cin=%d cout=%d mode=conv tile=%d k=1,1 s=1,1 p=0,0 d=1,1 stream=0\n"
          % (C, C, H, W, NB, Kc, Kc, H, W, NB, C, Kc, tile))
    for i in range(K):
        t += "copy src=s:0:%d:%d:%d:%d dst=d%d:0:%d:%d:%d:%d stream=1\n" % (CS, CS, H, W, i, CS, CS, H, W)
    plan = m.add_plan("p", t)
    m.write("x", x)
    m.write("s", src)
    plan.finalize()
    nbad = 0
    for rep in range(3):
        for i in range(K):
            m.write("d%d" % i, np.zeros_like(src))
        plan.run()
        ctx.sync()
        for i in range(K):
            d = m.read("d%d" % i, src.shape)
            if not np.array_equal(d, src):
                nbad += 1
                px = np.nonzero((d != src).reshape(H * W, CS).any(axis=1))[0]
                ch = np.nonzero((d != src).reshape(H * W, CS).any(axis=0))[0]
                vals = d.reshape(H * W, CS)[px[0], ch[:4]]
                if nbad <= 3:
                    print("  bind %d rep %d copy %d: %d pixels differ (first %d), channels %d..%d (%d), got %s expected %s"
                          % (it, rep, i, len(px), px[0], ch[0], ch[-1], len(ch), vals, src.reshape(H * W, CS)[px[0], ch[:4]]), flush=True)
    print("bind %d: %d of %d copies wrong" % (it, nbad, 3 * K), flush=True)
    bad_runs += nbad > 0
    m.close()
    ctx.close()
print("tile %d %s: %d of %d bindings had wrong copies" % (tile, "graph" if graph else "eager", bad_runs, runs))
