"""How much of |logit - oracle| at the worst pixel of the headline binding is the ORACLE's own rounding of the flow field?

The flow field is the one input of the path whose rounding error is amplified (a bilinear tap that straddles the image border
changes by d x |feature| for a flow error of d pixels).  This script evaluates FlowNet-S once more in float64 (torch
CPU: conv2d / conv_transpose2d / avg_pool2d) = the exact flow to fp32 rounding, and reports for the non-key frame of the config-4
clip: the distance of the oracle's fp32 flow and of the HIP path's flow (8 clips per call, 1 clip per call) from it, and the
logit error of either against the oracle RE-RUN ON THE EXACT FLOW.

    python scripts/debug/flow_exact.py
"""
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
sys.path.insert(0, os.path.join(HERE, "..", "..", "tests"))


def flownet64(P, img_cur, img_ref):
    """oracle.graphs.flownet in float64"""
    import torch
    import torch.nn.functional as F
    t = lambda a: torch.from_numpy(np.asarray(a, np.float64))
    lk = lambda x: F.leaky_relu(x, 0.1)

    def conv(name, x, stride, pad):
        return F.conv2d(x, t(P[name + "_weight"]), t(P[name + "_bias"]), stride=stride, padding=pad)

    data = torch.cat([t(img_cur) / 255.0, t(img_ref) / 255.0], dim=1)
    x = F.avg_pool2d(data, 2, 2)
    r1 = lk(conv("flow_conv1", x, 2, 3))
    r2 = lk(conv("conv2", r1, 2, 2))
    r3 = lk(conv("conv3", r2, 2, 2))
    r4 = lk(conv("conv3_1", r3, 1, 1))
    r5 = lk(conv("conv4", r4, 2, 1))
    r6 = lk(conv("conv4_1", r5, 1, 1))
    r7 = lk(conv("conv5", r6, 2, 1))
    r8 = lk(conv("conv5_1", r7, 1, 1))
    r9 = lk(conv("conv6", r8, 2, 1))
    r10 = lk(conv("conv6_1", r9, 1, 1))

    def refine(feat_in, skip, pred_name, deconv_name, upflow_name):
        pred = conv(pred_name, feat_in, 1, 1)
        h, w = skip.shape[2:]
        dec = F.conv_transpose2d(feat_in, t(P[deconv_name + "_weight"]), t(P[deconv_name + "_bias"]), stride=2)
        dec = lk(dec[:, :, 1:1 + h, 1:1 + w])
        up = F.conv_transpose2d(pred, t(P[upflow_name + "_weight"]), t(P[upflow_name + "_bias"]), stride=2)
        up = up[:, :, 1:1 + h, 1:1 + w]
        return torch.cat([skip, dec, up], dim=1)

    c2 = refine(r10, r8, "Convolution1", "deconv5", "upsample_flow6to5")
    c3 = refine(c2, r6, "Convolution2", "deconv4", "upsample_flow5to4")
    c4 = refine(c3, r4, "Convolution3", "deconv3", "upsample_flow4to3")
    c5 = refine(c4, r2, "Convolution4", "deconv2", "upsample_flow3to2")
    c5 = F.avg_pool2d(c5, 2, 2)
    flow = conv("Convolution5", c5, 1, 1)
    return (flow * 2.5).numpy()


def hip_flow(pred, H, W, nb):
    plan, lw = pred.plan_for(H, W, nb)
    arena = plan.arena()
    for kind, args in lw.ops:
        if kind == "conv" and args["name"] == "Convolution5":
            v = args["out"]
            b = v.buf
            a = arena[b.off:b.off + b.nbytes].view(np.float32).reshape(b.N, b.H, b.W, b.Cs)[..., v.coff:v.coff + v.C]
            return np.ascontiguousarray(a.transpose(0, 3, 1, 2))
    raise KeyError("Convolution5")


def main():
    from accel_amd import demo, mx
    from accel_amd.config.config import config, update_config
    from accel_amd.core import tester
    from accel_amd.utils import image, synth
    from oracle import graphs as G
    update_config(os.path.join(HERE, "..", "..", "tests", "golden", "dff_deeplab_vid_demo.yaml"))
    H, W, B, interval = 1024, 2048, 8, 2
    os.environ["ACCEL_ARENA_NO_REUSE"] = "1"
    config.SCALES[0] = (H, W)
    arg, aux = synth.model_params("18", H, W, config)
    clips = [synth.make_clip(H, W, 3)[:interval]] + [synth.make_clip(H, W, interval, seed=4100 + b) for b in range(1, B)]
    per_clip = [demo.build_batches(c, config) for c in clips]
    P = dict(arg)
    P.update(aux)
    fr = [image.transform(f, config.network.PIXEL_MEANS).astype(np.float32) for f in clips[0]]
    t0 = time.time()
    key = G.key_forward(P, fr[0])
    cur = G.cur_forward(P, "18", fr[1], fr[0], key["res5c_relu_output"])
    print("oracle: %.1f s" % (time.time() - t0), flush=True)
    t0 = time.time()
    fx = flownet64(P, fr[1], fr[0])
    print("float64 FlowNet: %.1f s; |flow|max %.3f" % (time.time() - t0, np.abs(fx).max()), flush=True)
    fo = cur["_flow"]
    print("oracle fp32 flow vs exact: max %.3e  rms %.3e" % (np.abs(fo - fx).max(), np.sqrt(((fo - fx) ** 2).mean())))
    saved = G.flownet
    G.flownet = lambda P_, a, b: fx.astype(np.float32)
    try:
        curx = G.cur_forward(P, "18", fr[1], fr[0], key["res5c_relu_output"])
    finally:
        G.flownet = saved
    lo, lx = cur["correction_output"][0], curx["correction_output"][0]
    d = np.abs(lo - lx)
    c, y, x = np.unravel_index(int(np.argmax(d)), d.shape)
    print("oracle(own flow) vs oracle(exact flow): max %.3e at class %d (%d, %d); at (632, 2039): %.3e" % (d.max(), c, y, x, d[:, 632, 2039].max()), flush=True)
    fy, fxx = 632 // 16, 2039 // 16
    print("flow error of the oracle around feature pixel (%d, %d): %s" % (fy, fxx, (fo - fx)[0, :, fy - 1:fy + 2, fxx - 1:fxx + 1].ravel()))
    for nb in (B, 1):
        rb = demo.ClipRunner("18", config, arg, aux, (H, W), batch=nb)
        try:
            for t in range(interval):
                if nb == 1:
                    arrays = per_clip[0][t]
                else:
                    arrays = [mx.nd.array(np.concatenate([per_clip[b][t][i].asnumpy() for b in range(nb)], axis=0)) for i in range(2)]
                    arrays.append(mx.nd.array(np.zeros((nb, 2048, 1, 1), np.float32)))
                logits, labels = rb.step(t, arrays, interval)
            lg = logits.asnumpy()[0]
            fh = hip_flow(rb.cur_predictor, H, W, nb)[0:1]
            print("HIP batch %d flow vs exact: max %.3e rms %.3e; vs oracle: max %.3e" % (nb, np.abs(fh - fx).max(), np.sqrt(((fh - fx) ** 2).mean()), np.abs(fh - fo).max()))
            print("   flow error around the feature pixel: %s" % ((fh - fx)[0, :, fy - 1:fy + 2, fxx - 1:fxx + 1].ravel()))
            for nm, ref in (("oracle(own flow)", lo), ("oracle(exact flow)", lx)):
                d = np.abs(lg - ref)
                c, y, x = np.unravel_index(int(np.argmax(d)), d.shape)
                print("   HIP batch %d vs %s: max %.3e at class %d (%d, %d); at (632, 2039): %.3e" % (nb, nm, d.max(), c, y, x, d[:, 632, 2039].max()), flush=True)
        finally:
            tester.release_models()


if __name__ == "__main__":
    main()
