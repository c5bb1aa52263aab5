"""Functional CPU restatement of the Accel test-time graphs, written directly
from the reference's symbol files (not from accel_amd's graph builder, so the
two descriptions check one another).

TEST INFRASTRUCTURE ONLY / PARITY UNPINNED -- see oracle/accel_oracle.c.

`P` everywhere is a dict {mxnet parameter name: fp32 numpy array} holding both
arg and aux params (names never collide: demo.py:192-195 merges them the same
way).  All tensors are NCHW fp32.
"""
import numpy as np

from . import ops as O

EPS_DCN = 1e-5      # self.eps, resnet_v1_101_flownet_deeplab.py:23
EPS_PREACT = 2e-5   # residual_unit / resnet(), :50-75,108


# DeformableConvolution is discontinuous where a sampling position crosses the image border (ops.deform_border_taps).
# While RECORD is a list, every deformable layer appends (layer name, input stride in image pixels, (k, 3) array of the
# output pixels (n, oy, ox) that have a tap within BORDER_EPS of that discontinuity): the parity checks may tolerate a
# deviation ONLY around such a pixel (tests/parity_report.py).
RECORD = None
BORDER_EPS = 1e-4
# The warp is ill-conditioned (not discontinuous) where a bilinear tap straddles the image border (ops.warp_border_points).
# While RECORD_WARP is a list, cur_forward appends (stride in image pixels, (k, 3) array of those output pixels): the parity
# report states the error inside and outside their footprints separately; the tolerance is the same everywhere.
RECORD_WARP = None


# fp16-MFMA mode of the HIP path (plan option dtype=f16, BASELINE config 5): the operands of every convolution with
# Cin % 8 == 0 and more than 4 output channels are rounded to half precision by the loader (weights once, on the host),
# products and sums stay fp32.  With ROUND_F16 set the oracle evaluates exactly that function -- rounded operands, fp32
# arithmetic -- so the reduced-precision mode is checked against ITS OWN specification, not only against the fp32 result.
ROUND_F16 = False


def _h(a):
    return np.asarray(a, np.float32).astype(np.float16).astype(np.float32)


def _f16_layer(cin, cout):
    return ROUND_F16 and cin % 8 == 0 and cout > 4


# Half activation STORAGE of the fp16 mode (accel_amd.lower.Lowering.assign_storage): the FUSED output of a convolution -- after
# scale / shift, residual and activation -- is rounded to half when it is stored iff every op that reads it is a half-capable
# convolution (as data input or as residual) and the layer itself is one; every reader then sees the rounded value.  The set of
# such layers follows from the graph alone (tests/test_host_cpu.py pins it for the ResNet-50 / -101 trunks: every branch convolution
# of res2-res5 except the res5 `branch2a` outputs, which the deformable sampler reads, and the last block, which is the propagated
# feature; FlowNet conv3 / 4 / 5 / 6; the right head's fc6); the tests hand the names in.
# STORE_F16: None, or the set of layer names (the convolution node's name) whose output is stored as half.
STORE_F16 = None


def _st(name, y):
    return _h(y) if (ROUND_F16 and STORE_F16 and name in STORE_F16) else y


def _dcn(name, stride_px, x, off, w, stride, pad, dilate, dg):
    if RECORD is not None:
        pts = O.deform_border_taps(x.shape[2], x.shape[3], off, w.shape[2:], stride, pad, dilate, BORDER_EPS)
        if len(pts):
            RECORD.append((name, stride_px, pts))
    if _f16_layer(x.shape[1] * w.shape[2] * w.shape[3], w.shape[0]):
        # the HIP path samples in fp32 (dcn_cols) and rounds the sampled COLUMNS when the GEMM loads them
        K = w.shape[0]
        w2 = _h(w).reshape(K, -1)
        outs = []
        for n in range(x.shape[0]):
            col = _h(O.deform_im2col(x[n], off[n:n + 1], w.shape[2:], stride, pad, dilate, dg))
            outs.append((w2 @ col).reshape(K, off.shape[2], off.shape[3]))
        return np.stack(outs).astype(np.float32)
    return O.deform_conv2d(x, off, w, stride=stride, pad=pad, dilate=dilate, dg=dg)


def _bn(P, name, x, eps, fix_gamma=False):
    return O.batchnorm(x, P[name + "_gamma"], P[name + "_beta"],
                       P[name + "_moving_mean"], P[name + "_moving_var"], eps, fix_gamma)


def _conv(P, name, x, stride=1, pad=0, dilate=1, bias=False, wname=None, bname=None):
    w = P[wname or name + "_weight"]
    b = P[bname or name + "_bias"] if bias else None
    if _f16_layer(w.shape[1], w.shape[0]):
        x, w = _h(x), _h(w)
    return O.conv2d(x, w, b, stride, pad, dilate)


# --------------------------------------------------------------------------
# ResNet-101 / ResNet-50 with deformable res5
# resnet_v1_101_flownet_deeplab.py:576-1300 (get_resnet_dcn), :235-574 (get_resnet_dcn_50)
# --------------------------------------------------------------------------
def resnet_dcn_bottleneck(P, data, units, prefix, unit_namer, dcn_dg, offset_dilated):
    p = prefix
    x = _conv(P, p + "conv1", data, stride=2, pad=3)
    x = O.relu(_bn(P, p + "bn_conv1", x, EPS_DCN))
    x = O.pool2d(x, "max", 3, 2, 0, "full")
    mids = [64, 128, 256, 512]
    for si, n in enumerate(units):
        stage = si + 2
        for ui, suffix in enumerate(unit_namer(stage, n)):
            u = "%d%s" % (stage, suffix)
            first = ui == 0
            # stride 2 sits on the 1x1 convs of res3a / res4a (:647-652,:734-739); res5 keeps stride 1
            s = 2 if (first and stage in (3, 4)) else 1
            if first:
                sc = _st(p + "res" + u + "_branch1", _bn(P, p + "bn" + u + "_branch1",
                                                         _conv(P, p + "res" + u + "_branch1", x, stride=s), EPS_DCN))
            else:
                sc = x
            y = _conv(P, p + "res" + u + "_branch2a", x, stride=s)
            y = _st(p + "res" + u + "_branch2a", O.relu(_bn(P, p + "bn" + u + "_branch2a", y, EPS_DCN)))
            if stage == 5:
                oname = p + "res" + u + "_branch2b_offset"
                if offset_dilated:   # 72-ch, pad 2 dilate 2 (:514-515)
                    off = _conv(P, oname, y, pad=2, dilate=2, bias=True)
                else:                # 18-ch, pad 1 (:1232-1234)
                    off = _conv(P, oname, y, pad=1, bias=True)
                y = _dcn(p + "res" + u + "_branch2b", 16, y, off, P[p + "res" + u + "_branch2b_weight"], 1, 2, 2, dcn_dg)
            else:
                y = _conv(P, p + "res" + u + "_branch2b", y, pad=1)
            y = _st(p + "res" + u + "_branch2b", O.relu(_bn(P, p + "bn" + u + "_branch2b", y, EPS_DCN)))
            y = _bn(P, p + "bn" + u + "_branch2c", _conv(P, p + "res" + u + "_branch2c", y), EPS_DCN)
            x = _st(p + "res" + u + "_branch2c", O.relu(sc + y))
    return x


def _namer101(stage, n):
    if stage in (3, 4):
        return ["a"] + ["b%d" % i for i in range(1, n)]
    return [chr(ord("a") + i) for i in range(n)]


def _namer50(stage, n):
    return [chr(ord("a") + i) for i in range(n)]


def resnet_dcn_101(P, data):
    return resnet_dcn_bottleneck(P, data, (3, 4, 23, 3), "", _namer101, 1, False)


def resnet_dcn_50(P, data):
    return resnet_dcn_bottleneck(P, data, (3, 4, 6, 3), "50_", _namer50, 4, True)


# --------------------------------------------------------------------------
# Pre-activation ResNet-18/34 trunk: resnet() :88-130 with residual_unit :71-86
# --------------------------------------------------------------------------
def resnet_preact_trunk(P, data, prefix, units):
    p = prefix
    x = _bn(P, p + "bn_data", data, EPS_PREACT, fix_gamma=True)
    x = _conv(P, p + "conv0", x, stride=2, pad=3)
    x = O.relu(_bn(P, p + "bn0", x, EPS_PREACT))
    x = O.pool2d(x, "max", 3, 2, 1, "valid")
    for i, n in enumerate(units):
        for j in range(n):
            name = "%sstage%d_unit%d" % (p, i + 1, j + 1)
            stride = (1 if i == 0 else 2) if j == 0 else 1
            act1 = O.relu(_bn(P, name + "_bn1", x, EPS_PREACT))
            c1 = _conv(P, name + "_conv1", act1, stride=stride, pad=1)
            act2 = O.relu(_bn(P, name + "_bn2", c1, EPS_PREACT))
            c2 = _conv(P, name + "_conv2", act2, stride=1, pad=1)
            if j == 0:   # dim_match=False on the first unit of EVERY stage (:122)
                sc = _conv(P, name + "_sc", act1, stride=stride)
            else:
                sc = x
            x = c2 + sc
    return x


def resnet_dcn_conv5_basic(P, feat, prefix, n_units):
    """get_resnet_dcn_18_conv5 :132-170 (2 units) / get_resnet_dcn_34_conv5 :172-233 (3)."""
    p = prefix
    x = feat
    for ui in range(n_units):
        u = "5" + chr(ord("a") + ui)
        if ui == 0:
            sc = _bn(P, p + "bn" + u + "_branch1", _conv(P, p + "res" + u + "_branch1", x, stride=2), EPS_DCN)
            y = _conv(P, p + "res" + u + "_branch2a", x, stride=2, pad=1)
        else:
            sc = x
            y = _conv(P, p + "res" + u + "_branch2a", x, stride=1, pad=1)
        y = O.relu(_bn(P, p + "bn" + u + "_branch2a", y, EPS_DCN))
        off = _conv(P, p + "res" + u + "_branch2b_offset", y, pad=2, dilate=2, bias=True)
        y = _dcn(p + "res" + u + "_branch2b", 32, y, off, P[p + "res" + u + "_branch2b_weight"], 1, 2, 2, 4)
        y = _bn(P, p + "bn" + u + "_branch2b", y, EPS_DCN)
        x = O.relu(sc + y)
    return x


# --------------------------------------------------------------------------
# FlowNet-S: get_flownet :1751-1808 (Convolution5_scale is dead on this path)
# --------------------------------------------------------------------------
def flownet(P, img_cur, img_ref):
    lk = O.leaky_relu
    data = np.concatenate([img_cur / np.float32(255.0), img_ref / np.float32(255.0)], axis=1)
    x = O.pool2d(data, "avg", 2, 2, 0, "full")
    r1 = lk(_conv(P, "flow_conv1", x, 2, 3, bias=True))
    r2 = lk(_conv(P, "conv2", r1, 2, 2, bias=True))
    r3 = _st("conv3", lk(_conv(P, "conv3", r2, 2, 2, bias=True)))
    r4 = lk(_conv(P, "conv3_1", r3, 1, 1, bias=True))
    r5 = _st("conv4", lk(_conv(P, "conv4", r4, 2, 1, bias=True)))
    r6 = lk(_conv(P, "conv4_1", r5, 1, 1, bias=True))
    r7 = _st("conv5", lk(_conv(P, "conv5", r6, 2, 1, bias=True)))
    r8 = lk(_conv(P, "conv5_1", r7, 1, 1, bias=True))
    r9 = _st("conv6", lk(_conv(P, "conv6", r8, 2, 1, bias=True)))
    r10 = lk(_conv(P, "conv6_1", r9, 1, 1, bias=True))

    def refine(feat_in, skip, pred_name, deconv_name, upflow_name):
        pred = _conv(P, pred_name, feat_in, 1, 1, bias=True)
        wd = P[deconv_name + "_weight"]
        rnd = _f16_layer(wd.shape[0], wd.shape[1])
        dec = O.deconv2d(_h(feat_in) if rnd else feat_in, _h(wd) if rnd else wd, P[deconv_name + "_bias"], 2, 0)
        dec = lk(O.crop_like(dec, skip.shape[2:], (1, 1)))
        up = O.deconv2d(pred, P[upflow_name + "_weight"], P[upflow_name + "_bias"], 2, 0)
        up = O.crop_like(up, skip.shape[2:], (1, 1))
        return np.concatenate([skip, dec, up], axis=1)

    c2 = refine(r10, r8, "Convolution1", "deconv5", "upsample_flow6to5")
    c3 = refine(c2, r6, "Convolution2", "deconv4", "upsample_flow5to4")
    c4 = refine(c3, r4, "Convolution3", "deconv3", "upsample_flow4to3")
    c5 = refine(c4, r2, "Convolution4", "deconv2", "upsample_flow3to2")
    c5 = O.pool2d(c5, "avg", 2, 2, 0, "full")
    flow = _conv(P, "Convolution5", c5, 1, 1, bias=True)
    return flow * np.float32(2.5)


# --------------------------------------------------------------------------
# task head: fc6 -> relu -> score -> Deconvolution 32x32 s16 g19 -> Crop(8,8)
# accel_18.py:136-155,178-197,208-227
# --------------------------------------------------------------------------
def head(P, feat, data_hw, prefix=""):
    p = prefix
    x = _st(p + "fc6", O.relu(_conv(P, p + "fc6", feat, bias=True)))
    s = _conv(P, p + "score", x, bias=True)
    ncls = s.shape[1]
    up = O.deconv2d(s, P[p + "upsampling_weight"], None, 16, 0, groups=ncls)
    return O.crop_like(up, data_hw, (8, 8))


def key_forward(P, data):
    """get_key_test_symbol (accel_18.py:121-159; identical in 34/50/101).
    returns {'res5c_relu_output', 'croped_score_output'}."""
    feat = resnet_dcn_101(P, data)
    score = head(P, feat, data.shape[2:])
    return {"res5c_relu_output": feat, "croped_score_output": score}


def deeplab_forward(P, data):
    """deeplab/symbols/resnet_v1_101_deeplab_dcn.py:786-821 at test time: the key graph's
    croped_score through SoftmaxOutput(multi_output=True) = softmax over the class axis."""
    s = key_forward(P, data)["croped_score_output"]
    e = np.exp(s - s.max(axis=1, keepdims=True), dtype=np.float32)
    return {"softmax_output": (e / e.sum(axis=1, keepdims=True, dtype=np.float32)).astype(np.float32)}


def cur_forward(P, version, data, data_key, feat_key):
    """get_cur_test_symbol: accel_18.py:161-239, accel_34.py:161-239,
    accel_50.py:156-228, accel_101.py:144-193."""
    version = str(version)
    flow = flownet(P, data, data_key)
    if RECORD_WARP is not None:
        RECORD_WARP.append((data.shape[2] // flow.shape[2], O.warp_border_points(flow)))
    warped = O.flow_warp(feat_key, flow)
    out = {"warping_feat_output": warped, "_flow": flow}
    hw = data.shape[2:]
    if version == "dff":        # Deep Feature Flow only: propagate, no correction branch
        out["croped_score_output"] = head(P, warped, hw)
        return out
    if version == "101":
        cur = resnet_dcn_101(P, data)
        fused = _st("correction", _conv(P, "corr", np.concatenate([warped, cur], axis=1), bias=True))
        out["croped_score_output"] = head(P, fused, hw)
        return out
    left = head(P, warped, hw)
    if version in ("18", "34"):
        p = version + "_"
        units = [2, 2, 2] if version == "18" else [3, 4, 6]
        r = resnet_preact_trunk(P, data, p, units)
        r = resnet_dcn_conv5_basic(P, r, p, 2 if version == "18" else 3)
        wu = P[p + "feat_upsampling_weight"]
        r = O.deconv2d(_h(r), _h(wu), None, 2, 1) if _f16_layer(wu.shape[0], wu.shape[1]) else O.deconv2d(r, wu, None, 2, 1)
        right = head(P, r, hw, p)
    elif version == "50":
        r = resnet_dcn_50(P, data)
        right = head(P, r, hw, "curr_")
    else:
        raise ValueError(version)
    out["correction_output"] = O.conv2d(np.concatenate([left, right], axis=1),
                                        P["corr_weight"], P["corr_bias"])
    return out


def train_forward(P, version, data, data_ref):
    """get_train_symbol, forward only: accel_18.py:31-119 (34 / 50 alike), accel_101.py:31-102.
    data 1x3xHxW (the labelled frame), data_ref (KEY_INTERVAL-1)x3xHxW (key frame, then the intermediate frames).
    The key frame's feature is warped once per frame pair (ref1<-ref0, ref2<-ref1, .., data<-ref_last; all flows from
    one FlowNet batch), then corrected by the branch on `data`; SoftmaxOutput(multi_output) = softmax over classes."""
    version = str(version)
    n = data_ref.shape[0]
    hw = data.shape[2:]
    if version == "101":
        both = resnet_dcn_101(P, np.concatenate([data_ref[:1], data], axis=0))
        feat, feat_cur = both[:1], both[1:]
    else:
        feat = resnet_dcn_101(P, data_ref[:1])
    nxt = np.concatenate([data_ref[1:], data], axis=0)
    flow = flownet(P, nxt, data_ref)
    for i in range(n):
        feat = O.flow_warp(feat, flow[i:i + 1])
    if version == "101":
        fused = _conv(P, "corr", np.concatenate([feat, feat_cur], axis=1), bias=True)
        s = head(P, fused, hw)
    else:
        left = head(P, feat, hw)
        if version in ("18", "34"):
            p = version + "_"
            r = resnet_preact_trunk(P, data, p, [2, 2, 2] if version == "18" else [3, 4, 6])
            r = resnet_dcn_conv5_basic(P, r, p, 2 if version == "18" else 3)
            r = O.deconv2d(r, P[p + "feat_upsampling_weight"], None, 2, 1)
            right = head(P, r, hw, p)
        elif version == "50":
            right = head(P, resnet_dcn_50(P, data), hw, "curr_")
        else:
            raise ValueError(version)
        s = O.conv2d(np.concatenate([left, right], axis=1), P["corr_weight"], P["corr_bias"])
    e = np.exp(s - s.max(axis=1, keepdims=True), dtype=np.float32)
    return {"softmax_output": (e / e.sum(axis=1, keepdims=True, dtype=np.float32)).astype(np.float32), "_logits": s, "_feat": feat}


def run_clip(P, version, frames, interval):
    """The demo.py:228-250 schedule on preprocessed frames (list of 1x3xHxW):
    idx % interval == 0 -> key graph, else cur graph with data_key = PREVIOUS
    frame and feat_key = previously propagated feature (demo.py:176-181,241).
    Returns per-frame (logits, labels)."""
    global RECORD, RECORD_WARP
    version = str(version)
    outs = ClipResult()
    feat = None
    prev = None
    carried = []       # discontinuity points of the key frame: they travel with the propagated feature
    carried_warp = []  # border-straddling warp pixels since the last key frame: the warped feature is the next frame's source
    for idx, im in enumerate(frames):
        if prev is None:
            prev = im
        saved, RECORD = RECORD, []
        saved_warp, RECORD_WARP = RECORD_WARP, []
        try:
            if idx % interval == 0:
                o = key_forward(P, im)
                feat = o["res5c_relu_output"]
                logits = o["croped_score_output"]
            else:
                o = cur_forward(P, version, im, prev, feat)
                feat = o["warping_feat_output"]
                logits = o["croped_score_output" if version in ("101", "dff") else "correction_output"]
            pts = [(name, int(n), (oy + 0.5) * s, (ox + 0.5) * s) for name, s, arr in RECORD for n, oy, ox in arr]
            wpts = [(int(n), (oy + 0.5) * s, (ox + 0.5) * s) for s, arr in RECORD_WARP for n, oy, ox in arr]
        finally:
            RECORD = saved
            RECORD_WARP = saved_warp
        if idx % interval == 0:
            carried = list(pts)
            carried_warp = []
            outs.critical.append(list(pts))
        else:
            outs.critical.append(carried + pts)
            carried_warp = carried_warp + wpts
        outs.warp_border.append(list(carried_warp))
        prev = im
        outs.append((logits, O.argmax_c(logits)))
    return outs


class ClipResult(list):
    """run_clip's result: a list of per-frame (logits, labels) that also carries, per frame, the image positions
    (layer, image index, y, x) of the deformable-convolution pixels that have a tap within BORDER_EPS of the border
    discontinuity -- in this frame's own layers and, on non-key frames, in the key frame whose feature they propagate."""

    def __init__(self, *a):
        super().__init__(*a)
        self.critical = []
        self.warp_border = []      # per frame: (image index, y, x) in image pixels, ops.warp_border_points of this and the earlier non-key frames
