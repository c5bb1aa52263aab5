"""Generates the committed golden fixtures from the CPU oracle.

PARITY UNPINNED: the reference holds no golden vectors for this path and its
arithmetic (MXNet) cannot run here, so these vectors pin the
ORACLE (and through it the HIP path), not the reference.  They are data:
seeded inputs and expected outputs.  Re-run with
    python tests/golden/make_golden.py
after any deliberate change of oracle semantics.
"""
import hashlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))

from accel_amd.config.config import config, reset_config, update_config  # noqa: E402
from accel_amd.utils import image, synth  # noqa: E402
from oracle import graphs as G, ops as O  # noqa: E402


def rnd(seed, *shape, scale=1.0):
    return (np.random.default_rng(seed).standard_normal(shape) * scale).astype(np.float32)


def op_vectors():
    d = {}
    x, w, b = rnd(1, 1, 8, 9, 11), rnd(2, 12, 8, 3, 3, scale=0.2), rnd(3, 12)
    d["conv3x3_s1_x"], d["conv3x3_s1_w"], d["conv3x3_s1_b"] = x, w, b
    d["conv3x3_s1_y"] = O.conv2d(x, w, b, 1, 1, 1)
    d["conv3x3_s2_y"] = O.conv2d(x, w, b, 2, 1, 1)
    d["conv3x3_d2_y"] = O.conv2d(x, w, b, 1, 2, 2)
    w7 = rnd(4, 6, 8, 7, 7, scale=0.1)
    d["conv7x7_w"], d["conv7x7_s2_y"] = w7, O.conv2d(x, w7, None, 2, 3, 1)
    wd, bd = rnd(5, 8, 5, 4, 4, scale=0.2), rnd(6, 5)
    d["deconv4_w"], d["deconv4_b"] = wd, bd
    d["deconv4_p0_y"] = O.deconv2d(x, wd, bd, 2, 0)
    d["deconv4_p1_y"] = O.deconv2d(x, wd, bd, 2, 1)
    s, wu = rnd(7, 1, 3, 2, 3), synth.bilinear_kernel(3, 32)
    d["up_s"], d["up_y"] = s, O.crop_like(O.deconv2d(s, wu, None, 16, 0, groups=3), (32, 48), (8, 8))
    xo = rnd(8, 1, 5, 7, 9)
    d["pool_x"] = xo
    d["pool_max_full_y"] = O.pool2d(xo, "max", 3, 2, 0, "full")
    d["pool_max_valid_p1_y"] = O.pool2d(xo, "max", 3, 2, 1, "valid")
    d["pool_avg_full_y"] = O.pool2d(xo, "avg", 2, 2, 0, "full")
    g, be, mu, var = rnd(9, 5) + 1.5, rnd(10, 5), rnd(11, 5), np.abs(rnd(12, 5)) + 0.5
    d["bn_g"], d["bn_b"], d["bn_m"], d["bn_v"] = g, be, mu, var
    d["bn_y"] = O.batchnorm(xo, g, be, mu, var, 1e-5)
    d["bn_fixg_y"] = O.batchnorm(xo, g, be, mu, var, 2e-5, fix_gamma=True)
    feat, flow = rnd(13, 1, 4, 6, 8), rnd(14, 1, 2, 6, 8, scale=2.5)
    d["warp_feat"], d["warp_flow"], d["warp_y"] = feat, flow, O.flow_warp(feat, flow)
    xd, wdc = rnd(15, 1, 16, 6, 7), rnd(16, 6, 16, 3, 3, scale=0.2)   # 4 channels per group at dg=4
    d["dcn_x"], d["dcn_w"] = xd, wdc
    for dg in (1, 4):
        off = rnd(17 + dg, 1, 18 * dg, 6, 7, scale=2.0)
        d["dcn_off_dg%d" % dg] = off
        d["dcn_y_dg%d" % dg] = O.deform_conv2d(xd, off, wdc, 1, 2, 2, dg)
    return d


def weights_sha(arg, aux):
    h = hashlib.sha256()
    for k in sorted(arg):
        h.update(k.encode())
        h.update(np.ascontiguousarray(arg[k]).tobytes())
    for k in sorted(aux):
        h.update(k.encode())
        h.update(np.ascontiguousarray(aux[k]).tobytes())
    return h.hexdigest()


def chain(version, H=128, W=256, n_frames=4, interval=3):
    arg, aux = synth.model_params(version, H, W, config)
    frames = synth.make_clip(H, W, n_frames)
    P = dict(arg)
    P.update(aux)
    outs = G.run_clip(P, version, [image.transform(f, config.network.PIXEL_MEANS).astype(np.float32) for f in frames],
                      interval)
    logits = np.stack([o[0][0] for o in outs])             # (T, 19, H, W)
    labels = np.stack([o[1][0] for o in outs])             # (T, H, W)
    return {"frames": np.stack(frames), "labels": labels, "logits_sub4": logits[:, :, ::4, ::4].copy(),
            "logits_absmax": np.abs(logits).max(axis=(1, 2, 3)),
            "logits_sum": logits.astype(np.float64).sum(axis=(1, 2, 3)),
            "interval": np.int32(interval), "weights_sha256": np.array(weights_sha(arg, aux))}


if __name__ == "__main__":
    reset_config()
    update_config(os.path.join(HERE, "dff_deeplab_vid_demo.yaml"))
    np.savez_compressed(os.path.join(HERE, "ops_golden.npz"), **op_vectors())
    for v in ("18", "101"):
        np.savez_compressed(os.path.join(HERE, "chain_accel%s_128x256.npz" % v), **chain(v))
    print("golden fixtures written")
