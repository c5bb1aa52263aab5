"""Where does a non-finite range come from?  After every run: every slot's partial words; for a non-finite one, the reader's input tensor
(arena kept private: ACCEL_ARENA_NO_REUSE=1) is scanned for non-finite values."""
import sys, os, ctypes
os.environ.setdefault("ACCEL_ARENA_NO_REUSE", "1")
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from accel_amd import demo, mx, runtime
from accel_amd.config.config import config, update_config
from accel_amd.core import tester
from accel_amd.utils import image, synth

H, W = 256, 512
update_config(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests", "golden", "dff_deeplab_vid_demo.yaml"))
config.SCALES[0] = (H, W)
arg, aux = synth.model_params("18", H, W, config)
A = synth.make_clip(H, W, 2, seed=5101)
pre = lambda fr, f: [(np.float32(f) * image.transform(im, config.network.PIXEL_MEANS)).astype(np.float32) for im in fr]
seqs = [pre(A, 1.0), pre([np.zeros_like(A[0])] * 2, 1.0), pre(A, 8.0), pre(A, 0.01), pre(A, 1.0), pre(A, 100.0), pre(A, 1e-3)]
lib = runtime.lib()


def scan(plan, lw, tag):
    ops = plan.ops()
    bad = 0
    for i, o in enumerate(ops):
        if o["kind"] != "conv":
            continue
        w = np.zeros(1088, np.uint32)
        if lib.accel_plan_op_range_words(plan.handle, i, w.ctypes.data_as(ctypes.c_void_p), 1088) != 0:
            continue
        nf = np.nonzero(w >= 0x7F800000)[0]
        if len(nf):
            bad += 1
            args = [a for k, a in lw.ops if k == "conv" and a.get("name") == o["name"]][0]
            v = args["in"]
            arena = plan.arena()
            b = v.buf
            t = arena[b.off:b.off + b.nbytes].view(np.float32).reshape(b.N, b.H, b.W, b.Cs)[..., v.coff:v.coff + v.C] if b.space == "A" else None
            print("%s: conv %s (tile %d) slot words %s = %s; input view %s: non-finite values %s, max |x| %s" % (
                tag, o["name"], o["tile"], nf[:8], [hex(int(x)) for x in w[nf[:8]]], v.ref(),
                None if t is None else int((~np.isfinite(t)).sum()), None if t is None else float(np.abs(t[np.isfinite(t)]).max())))
    return bad


for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 3):
    r = demo.ClipRunner("18", config, arg, aux, (H, W))
    zero = mx.nd.array(np.zeros((1, 2048, 1, 1), np.float32))
    for si, frames in enumerate(seqs):
        prev = None
        for idx, im in enumerate(frames):
            cur = mx.nd.array(im)
            prev = prev or cur
            try:
                r.step(idx, [cur, prev, zero], 2)
            except runtime.AccelError as e:
                print("rep %d seq %d frame %d: %s" % (rep, si, idx, str(e)[:160]))
                r.step(idx, [cur, prev, zero], 2)
            pred = r.key_predictor if idx == 0 else r.cur_predictor
            plan, lw = pred.plan_for(H, W, 1)
            scan(plan, lw, "rep %d seq %d frame %d" % (rep, si, idx))
            prev = cur
    tester.release_models()
print("done")
