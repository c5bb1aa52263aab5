"""Two-stream diagnosis: bind the Accel non-key plan N times with two streams; after the first non-key frame of every
binding compare (a) the logits of the captured two-stream replay, (b) the logits of the SAME bound plan run op by op on
one stream (accel_plan_run_serial), (c) a one-stream binding of the same process.  When (a) != (b) the arena of both
runs is diffed op by op and the first ops whose outputs depend on the schedule are named.

    ACCEL_FOLD_LINEAR=0 python scripts/debug/twostream_bisect.py [N] [version] [HxW]
"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np

from accel_amd import demo
from accel_amd.config.config import config, update_config
from accel_amd.core import tester
from accel_amd.lower import View
from accel_amd.utils import synth

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
update_config(os.path.join(ROOT, "tests", "golden", "dff_deeplab_vid_demo.yaml"))
N = int(sys.argv[1]) if len(sys.argv) > 1 else 12
version = sys.argv[2] if len(sys.argv) > 2 else "18"
H, W = [int(v) for v in (sys.argv[3] if len(sys.argv) > 3 else "1024x2048").split("x")]
config.SCALES[0] = (H, W)
arg, aux = synth.model_params(version, H, W, config)
frames = synth.make_clip(H, W, 2)
data = demo.build_batches(frames, config)


def one_binding(multi):
    os.environ["ACCEL_MULTI_STREAM"] = "1" if multi else "0"
    r = demo.ClipRunner(version, config, arg, aux, (H, W))
    r.step(0, data[0], 5)[0].asnumpy()
    lg = r.step(1, data[1], 5)[0].asnumpy()
    return r, lg


def op_views(lw):
    out = []
    for i, (kind, args) in enumerate(lw.ops):
        for k in ("dst", "out", "out2"):
            v = args.get(k)
            if isinstance(v, View) and v.buf.space == "A":
                out.append((i, kind, args.get("name", ""), args.get("stream", 0), k, v))
    return out


def region(arena, v):
    b = v.buf
    a = arena[b.off:b.off + b.nbytes].view(np.float32).reshape(b.N * b.H * b.W, b.Cs)
    px = b.H * b.W
    return a[v.img0 * px:(v.img0 + v.nimg) * px, v.coff:v.coff + v.C]


def probe_records(what):
    """diagnostics build (-DACCEL_ORDER_PROBE): did dcn_cols ever see a split-K reduce wavefront of its own stream in flight?"""
    import ctypes
    from accel_amd import runtime
    lib = runtime.lib()
    if not hasattr(lib, "accel_debug_probe"):
        return
    buf = (ctypes.c_uint32 * 16)()
    lib.accel_debug_probe(buf, 1)
    for st in range(2):
        q = buf[8 * st: 8 * st + 8]
        if q[4]:
            print("    order probe, %s, stream %d: reduce wavefronts started %d finished %d; dcn_cols wavefronts %d, of which %d saw reduce "
                  "wavefronts of this stream in flight (max %d)" % (what, st, q[0], q[1], q[4], q[2], q[3]), flush=True)


def dcn_records(what):
    probe_records(what)
    """diagnostics build (-DACCEL_DCN_CHECK): corner fetches of dcn_cols that a second, cache-bypassing fetch contradicted"""
    import ctypes
    from accel_amd import runtime
    lib = runtime.lib()
    if not hasattr(lib, "accel_debug_dcn"):
        return
    buf = (ctypes.c_uint32 * (16 + 256 * 16))()
    lib.accel_debug_dcn(buf, len(buf), 1)
    n = buf[0]
    if n:
        print("    dcn_cols self-check, %s: %d corner fetches contradicted by the re-fetch" % (what, n))
    a = np.frombuffer(buf, np.uint32)
    for sl in range(min(n, 12)):
        d = a[16 + sl * 16: 32 + sl * 16]
        f = d[5:9].view(np.float32)
        print("      pix %d tap %d c4 %d lane %d corner %d: vector load (%.4f, %.4f) re-fetch (%.4f, %.4f); block %d hw_id %08x xcc %d"
              % (d[0], d[1], d[2], d[3], d[4], f[0], f[1], f[2], f[3], d[11], d[12], d[13] & 0xF), flush=True)


r, ref = one_binding(False)
dcn_records("one-stream reference")
r.close()
bad = []
for it in range(N):
    r, lg = one_binding(True)
    plan, lw = r.cur_predictor.plan_for(H, W, 1)
    m = r.cur_predictor._model
    a_graph = plan.arena()
    dcn_records("two-stream run")
    plan.run_serial()
    lg_serial = m.read("logits", (1, 19, H, W))
    dcn_records("serial run")
    e_gs = float(np.abs(lg - lg_serial).max())
    e_gr = float(np.abs(lg - ref).max())
    e_sr = float(np.abs(lg_serial - ref).max())
    print("bind %2d: |graph-serial| %.3g  |graph-onestream ref| %.3g  |serial-ref| %.3g" % (it, e_gs, e_gr, e_sr), flush=True)
    if e_gs > 0:
        bad.append(it)
        a_serial = plan.arena()
        ops = plan.ops()
        shown = 0
        for i, kind, name, st, key, v in op_views(lw):
            d = np.abs(region(a_graph, v) - region(a_serial, v))
            if d.max() > 0:
                rows = np.nonzero(d.max(axis=1))[0]
                print("    op %3d %-10s %-40s stream %d %-4s tile %3d ksplit %d: max diff %.3g, %d of %d pixels (first %d last %d)"
                      % (i, kind, name, st, key, ops[i]["tile"], ops[i]["ksplit"], d.max(), len(rows), d.shape[0], rows[0], rows[-1]), flush=True)
                if shown == 0 and os.environ.get("ACCEL_ARENA_NO_REUSE") == "1":
                    b = v.buf
                    dm = d.max(axis=1).reshape(b.N * b.H, b.W)
                    rband = (dm > 0).reshape(8, -1, b.W).sum(axis=(1, 2))
                    cband = (dm > 0).reshape(b.N * b.H, 8, -1).sum(axis=(0, 2))
                    print("        differing pixels by row band (8 bands): %s" % rband.tolist())
                    print("        differing pixels by column band (8 bands): %s" % cband.tolist(), flush=True)
                if shown == 0 and kind == "dcn_cols" and os.environ.get("ACCEL_ARENA_NO_REUSE") == "1":
                    args = lw.ops[i][1]
                    offv, xin = args["off"], args["in"]
                    off_g, off_s = region(a_graph, offv), region(a_serial, offv)
                    x_g, x_s = region(a_graph, xin), region(a_serial, xin)
                    print("        inputs at the end of the runs: |off graph - serial| %.3g, |x graph - serial| %.3g"
                          % (np.abs(off_g - off_s).max(), np.abs(x_g - x_s).max()))
                    C = xin.C
                    dg = int(args["dg"])
                    g_reg, s_reg = region(a_graph, v), region(a_serial, v)
                    for px in rows[:4]:
                        dd = d[px].reshape(9, C)
                        taps = np.nonzero(dd.max(axis=1))[0]
                        for tp in taps[:3]:
                            ch = np.nonzero(dd[tp])[0]
                            gv, sv = g_reg[px].reshape(9, C)[tp], s_reg[px].reshape(9, C)[tp]
                            grp = ch[0] // (C // dg)
                            o = off_s[px, grp * 18 + 2 * tp: grp * 18 + 2 * tp + 2]
                            print("        pixel %d (y %d x %d) tap %d: %d channels differ (ch %d..%d), group %d offset (%.6f, %.6f); graph %s serial %s"
                                  % (px, px // v.buf.W, px % v.buf.W, tp, len(ch), ch[0], ch[-1], grp, o[0], o[1],
                                     np.array2string(gv[ch[:3]], precision=4), np.array2string(sv[ch[:3]], precision=4)), flush=True)
                shown += 1
                if shown >= 6:
                    break
    r.close()
print("%d of %d two-stream bindings: graph replay differs from the serial run of the same binding: %s (version %s, b3=%s fold=%s graph=%s)"
      % (len(bad), N, bad, version, os.environ.get("ACCEL_BF16X3", "1"), os.environ.get("ACCEL_FOLD_LINEAR", "1"), os.environ.get("ACCEL_HIP_GRAPH", "1")))
