"""Lowering: MXNet-style symbol graph -> fused HIP kernel plan (text).

This is where the MI355X design lives.  The reference hands its graph to
MXNet, which runs one library kernel per operator over NCHW tensors (several
hundred launches and ~3 GB of activation traffic per key frame).
Here the graph is rewritten for the hardware before anything runs:

  * activations are NHWC (channel stride padded to 4 floats) so every conv is an
    implicit GEMM whose K axis is contiguous (16-byte loads, MFMA operands);
  * BatchNorm (folded to scale/shift), bias, scalar multiply, residual add,
    ReLU / LeakyReLU and the pre-activation BN+ReLU of the NEXT unit are fused
    into the producing conv's epilogue (conv_igemm.hip);
  * Concat never materialises: producers write straight into channel slices of
    the concat buffer;
  * Deconvolution 4x4/2 (+Crop offset 1) becomes four sub-pixel 2x2 convs;
  * image/255 + Concat + avg-pool -> one `prep_flow` kernel; bn_data -> the
    NCHW->NHWC4 image converter;
  * GridGenerator + BilinearSampler -> one `warp` kernel;
  * Deconvolution 32x32/16 + Crop + Concat + correction 1x1 + argmax -> one
    `score_tail` kernel writing NCHW logits and the uint8 label map;
  * activation buffers get static offsets in one arena from a liveness plan,
    and the whole op list is captured into a hipGraph by the executor.

The output is the plan text consumed by accel_model_add_plan()
(accel_amd/csrc/accel_hip.cpp documents the line format).
"""
import numpy as np

from .mx.symbol import infer_shapes

ALIGN = 256


def _r4(c):
    return (c + 3) // 4 * 4


def _forced_tiles():
    """diagnostics: ACCEL_FORCE_TILE="18_conv0=3,fc6=74" pins the launch geometry of the named convolutions"""
    import os
    out = {}
    for tok in os.environ.get("ACCEL_FORCE_TILE", "").split(","):
        if "=" in tok:
            k, v = tok.split("=", 1)
            out[k.strip()] = int(v)
    return out


class VBuf(object):
    """A physical buffer: arena slot (offset assigned later) or persistent."""

    def __init__(self, bid, Cs, H, W, space="A", N=1):
        self.id, self.Cs, self.H, self.W, self.space, self.N = bid, Cs, H, W, space, N
        self.first, self.last = None, None
        self.off = 0
        self.esize = 4      # bytes per element: 4 = fp32; 2 = half (f16-mode plans with half activation storage, assign_storage)

    @property
    def nbytes(self):
        return self.N * self.H * self.W * self.Cs * self.esize

    def touch(self, op_idx):
        if self.first is None:
            self.first = op_idx
        self.last = op_idx


class View(object):
    """Channel slice [coff, coff + C) of images [img0, img0 + nimg) of a physical buffer (images are stacked pixel-major,
    so a batch slice is a contiguous sub-range)."""

    def __init__(self, buf, C, coff=0, img0=0, nimg=None):
        self.buf, self.C, self.coff, self.img0 = buf, C, coff, img0
        self.nimg = buf.N if nimg is None else nimg

    def image(self, i0, n):
        """images [i0, i0 + n) of this view (mx.sym.split along the batch axis)"""
        assert 0 <= i0 and i0 + n <= self.nimg
        return View(self.buf, self.C, self.coff, self.img0 + i0, n)

    @property
    def H(self):
        return self.buf.H

    @property
    def W(self):
        return self.buf.W

    def ref(self):
        es = self.buf.esize
        off = self.buf.off + self.coff * es + self.img0 * self.buf.H * self.buf.W * self.buf.Cs * es
        r = "%s:%d:%d:%d:%d:%d" % (self.buf.space, off, self.C, self.buf.Cs, self.buf.H, self.buf.W)
        if es == 2:
            return r + ":%d:h" % self.nimg                        # half storage: the tag follows an explicit batch size
        return r if self.nimg == 1 else r + ":%d" % self.nimg     # batch: images stacked pixel-major


class Lowering(object):
    def __init__(self, sym, input_shapes, ncls=19, fold_linear=True, feat_slot=None, store_f16=False, fold_fusion=True):
        self.sym = sym
        self.store_f16 = bool(store_f16)      # f16-mode plans: arena buffers between capable convolutions are kept as half
        self.half_bufs = []
        self.fold_linear = bool(fold_linear)
        self.fold_fusion = bool(fold_linear) and bool(fold_fusion)      # the split of a fusion over Concat(warp, other): fp32-class plans only (lower())
        # Ping-pong of the propagated feature (non-key graphs): a warp cannot run in place, so the plan either warps into
        # scratch and copies back (feat_slot None: two copies of 0.8 GB per call of 8 frames), or exists TWICE: variant 0
        # reads `feat` / `featG` and writes `feat_b` / `featG_b`, variant 1 the other way round, and the host runs the
        # variant whose source holds the current feature.  Key plans always write `feat` / `featG`.
        self.feat_slot = feat_slot
        self.feat_src = "feat_b" if feat_slot == 1 else "feat"
        self.feat_dst = None if feat_slot is None else ("feat" if feat_slot == 1 else "feat_b")
        self.derived = {}      # derived parameter name -> ("deconv4x4s2*conv1x1", deconv weight, conv weight)
        self.derived_bufs = {} # derived persistent buffer -> {"from", "w", "cin", "cout", "H", "W"} (see lower_warp)
        # Feature fusion over Concat(warped feature, other) (Accel-101, accel_101.py:164-169): conv id -> (Concat node, warp node, other);
        # warp id -> conv; conv id -> view of the warped image of W_left * feat (plan_concats / lower_warp / lower_anchor)
        self.split_fusion, self.fusion_of_warp, self.fusion_res = {}, {}, {}
        self.shapes = infer_shapes(sym, input_shapes)
        self.nodes = sym.topo()
        # Accuracy budget: the flow field is the one place where fp32 rounding is AMPLIFIED -- a flow error of
        # d pixels moves the bilinear tap by d, and where the tap straddles the image border the warped feature changes by
        # d x |feature| (hundreds), i.e. 1e-5 px is 1e-3 on a logit.  Every layer that feeds the flow input of a warp therefore
        # stays off the Winograd-on-bf16 launch geometries (41 / 42; about 3x the rounding error of a direct evaluation):
        # measured at 1024x2048, non-key frame, pixel (632, 2039): 1.0e-3 with them, 4.0e-4 without (scripts/debug/wb3_budget.py).
        self.flow_ancestors = set()
        stack = []
        for n_ in self.nodes:
            if n_.op == "GridGenerator":
                stack.append(n_.inputs[0])
            elif n_.op == "Custom" and len(n_.inputs) == 2:
                stack.append(n_.inputs[1])
        while stack:
            m_ = stack.pop()
            if id(m_) in self.flow_ancestors:
                continue
            self.flow_ancestors.add(id(m_))
            stack.extend(m_.inputs)
        # The `offset` input of a DeformableConvolution is a sampling position as well.  Its PRODUCER alone is tagged (the offset
        # branch is one 3x3 convolution; the layers in front of it also feed the data path, where rounding is not amplified):
        # measured in round 6 (scripts/debug/layer_diff.py, profiles/r06_margin_bisect.log) the offsets of two launch-geometry
        # tables differ by 1.3e-6 of their range and the sampled columns by 1.6e-6 -- ordinary summation-order noise -- so the tag
        # changes no decision today; it keeps a future Winograd form away from them.
        for n_ in self.nodes:
            if n_.op == "DeformableConvolution" and len(n_.inputs) > 1:
                self.flow_ancestors.add(id(n_.inputs[1]))
        self.heads = sym._heads()
        self.head_ids = set(id(h) for h in self.heads)
        self.ops = []          # (kind, dict, [views read], [views written])
        self.bufs = []
        self.pbufs = {}        # name -> bytes
        self.val = {}          # id(node) -> View
        self.absorbed = set()
        self.concat_slot = {}  # id(node) -> View inside a concat buffer
        self.outputs = {}      # output name -> descriptor
        self.ncls = ncls
        self.total_flops = 0.0
        self.cons = {}
        for n in self.nodes:
            for idx, i in enumerate(n.inputs):
                if n.op == "Crop" and idx == 1:
                    continue   # shape reference only
                self.cons.setdefault(id(i), []).append(n)
        data_shape = input_shapes["data"]
        self.N, self.H, self.W = int(data_shape[0]), int(data_shape[2]), int(data_shape[3])
        self.nsfx = "" if self.N == 1 else ":%d" % self.N     # batch suffix of literal buffer references
        # image inputs (N x 3 x H x W variables): `data`, `data_key` of the test graphs, `data_ref` of the training graphs
        self.image_vars = {}
        for n in self.nodes:
            if n.op == "null" and n.name in input_shapes and len(input_shapes[n.name]) == 4 and int(input_shapes[n.name][1]) == 3:
                self.image_vars[n.name] = int(input_shapes[n.name][0])
        for name in ("data", "data_key"):
            self.image_vars.setdefault(name, self.N)
        for name, nimg in self.image_vars.items():
            self.pbufs[name] = nimg * 3 * self.H * self.W * 4

    # ---- helpers ---------------------------------------------------------------
    def shape(self, n):
        return self.shapes[id(n)]

    def consumers(self, n):
        return self.cons.get(id(n), [])

    def new_buf(self, C, H, W, Cs=None, N=None):
        b = VBuf(len(self.bufs), Cs or _r4(C), H, W, N=self.N if N is None else N)
        self.bufs.append(b)
        return b

    def pbuf_view(self, name, C, H, W, Cs=None, N=None):
        Cs = Cs or _r4(C)
        N = self.N if N is None else N
        b = VBuf(-1, Cs, H, W, space=name, N=N)
        self.pbufs[name] = max(self.pbufs.get(name, 0), N * H * W * Cs * 4)
        return View(b, C)

    def _head_on_feature(self, conv, feature):
        """`conv` = 1x1/1 Convolution with bias on the propagated feature whose only consumer is a ReLU
        (the DeepLab head's fc6): returns that ReLU node, else None."""
        a = conv.attrs
        if not self.fold_linear or conv.op != "Convolution" or conv.inputs[0] is not feature:
            return None
        if a["kernel"] != (1, 1) or a["stride"] != (1, 1) or a["pad"] != (0, 0) or a["num_group"] != 1 or a["no_bias"]:
            return None
        if a["num_filter"] % 4 or id(conv) in self.head_ids:
            return None
        cons = self.consumers(conv)
        if len(cons) != 1 or cons[0].op != "Activation" or cons[0].attrs["act_type"] != "relu":
            return None
        return cons[0]

    def _feat_image(self, conv, C, H, W, name="featG", wname=None, cin=None):
        """The derived persistent buffer `name` = W * feat (no bias), kept in step with `feat` by the plans: featG = fc6_weight * feat
        (the task head on the propagated feature), featC = the left half of corr_weight * feat (Accel-101's feature fusion)."""
        if cin is None:
            _, cin, _, _ = self.shape(conv.inputs[0])
        if self.feat_slot == 1:
            # the `_b` pair is only ever written by variant-0 plans, feature and image together: never stale
            return self.pbuf_view(name + "_b", C, H, W)
        v = self.pbuf_view(name, C, H, W)
        self.derived_bufs[name] = {"from": "feat", "w": wname or conv.inputs[1].name, "cin": cin, "cout": C, "H": H, "W": W, "N": self.N}
        return v

    def dest_for(self, node):
        """Where the value of `node` must be written."""
        if id(node) in self.concat_slot:
            return self.concat_slot[id(node)]
        N, C, H, W = self.shape(node)
        if id(node) in self.head_ids and node.name == "res5c_relu":
            return self.pbuf_view("feat", C, H, W, N=N)
        return View(self.new_buf(C, H, W, N=N), C)

    @staticmethod
    def _bkey(v):
        return ("A", v.buf.id) if v.buf.space == "A" else ("P", v.buf.space)

    def emit(self, kind, args, reads, writes, flops=0.0, nbytes=0.0, nbytes_fixed=0.0, n=None):
        idx = len(self.ops)
        for v in list(reads) + list(writes):
            if v is not None and v.buf.space == "A":
                v.buf.touch(idx)
        args = dict(args)
        # every plan runs on ONE stream, in list order (a two-stream lowering existed in rounds 1-3 and was removed: on this stack a
        # side-stream kernel could read stale granules of an earlier same-stream kernel's output while a bandwidth-heavy kernel ran
        # on the other hardware queue -- DESIGN.md 7)
        n = self.N if n is None else n
        force = _forced_tiles().get(args.get("name"))
        if force is not None and kind == "conv":
            args["tile"] = force       # diagnostics: ACCEL_FORCE_TILE="opname=tile,..."
        flops, nbytes = flops * n, nbytes * n + nbytes_fixed     # callers give per-image work (+ weights, read once)
        if flops:
            args["flops"] = "%.6g" % flops
            self.total_flops += flops
        if nbytes:
            args["bytes"] = "%.6g" % nbytes
        self.ops.append((kind, args))

    # ---- pre-pass: give Concat inputs their slices ---------------------------------
    def plan_concats(self):
        for n in self.nodes:
            if n.op != "Concat" or n.attrs.get("dim", 1) != 1:
                continue
            if all(i.op == "_div_scalar" for i in n.inputs):
                continue    # FlowNet input concat -> prep_flow
            if all(i.op == "Crop" and i.inputs[0].op == "Deconvolution" and
                   i.inputs[0].attrs["kernel"] == (32, 32) for i in n.inputs):
                continue    # score concat -> score_tail
            if self._plan_split_fusion(n):
                continue
            N, C, H, W = self.shape(n)
            buf = self.new_buf(C, H, W, N=N)
            off = 0
            for k, i in enumerate(n.inputs):
                ci = self.shape(i)[1]
                if off % 4:
                    raise NotImplementedError("Concat %s: slice %d starts at channel %d (not 16-byte aligned)" % (n.name, k, off))
                if id(i) in self.concat_slot:
                    raise NotImplementedError("%s feeds two Concats" % i.name)
                self.concat_slot[id(i)] = View(buf, ci, off)
                off += ci
            self.val[id(n)] = View(buf, C, 0)
            self.absorbed.add(id(n))

    def _plan_split_fusion(self, cat):
        """Linear fold (fold_linear): a 1x1 Convolution over Concat(warp(feat_key), other) -- Accel-101's feature fusion `correction`,
        accel_101.py:164-169 -- is  W_left * warp(F) + W_right * other + b,  and a 1x1 convolution commutes with the warp (lower_warp):
        W_left * warp(F) = warp(W_left * F).  The non-key plan therefore warps the 2048-channel image featC = W_left * feat (a derived
        persistent buffer like featG: rebuilt by the one-conv plan `init:featC` when a key frame has replaced `feat`, then carried from
        frame to frame by the warp itself) and runs the fusion over `other` alone with the warped image as its residual: half the
        multiply-adds of the reference's layer, no 4096-channel concat buffer, and the propagated feature can ping-pong between its two
        buffers instead of being copied back.  Same function; the summation order of the contraction changes (two halves)."""
        if not self.fold_fusion or len(cat.inputs) != 2:
            return False
        B, other = cat.inputs
        cons = self.consumers(cat)
        if len(cons) != 1 or cons[0].op != "Convolution" or cons[0].inputs[0] is not cat or id(cat) in self.head_ids:
            return False
        conv = cons[0]
        a = conv.attrs
        if a["kernel"] != (1, 1) or a["stride"] != (1, 1) or a["pad"] != (0, 0) or a["num_group"] != 1 or a["num_filter"] % 4:
            return False
        if B.op not in ("BilinearSampler", "Custom") or id(B) not in self.head_ids or len(self.consumers(B)) != 1:
            return False
        feat = B.inputs[0]
        if feat.op != "null" or feat.name != "feat_key" or self.shape(B)[1] % 4:
            return False
        self.split_fusion[id(conv)] = (cat, B, other)
        self.fusion_of_warp[id(B)] = conv
        self.absorbed.add(id(cat))
        return True

    # ---- image entry points ------------------------------------------------------------
    def image_sources(self, node):
        """`node` as a list of input images [(variable name, image index)], or None: an image variable, a batch slice
        of one (mx.sym.split axis 0) or a batch concatenation of those (mx.sym.Concat dim 0) -- how the training graphs
        assemble their FlowNet / ResNet inputs from `data` and `data_ref` (accel_18.py:43-48, accel_101.py:43-57)."""
        if node.op == "null":
            return [(node.name, i) for i in range(self.image_vars[node.name])] if node.name in self.image_vars else None
        if node.op == "_split_out":
            src = self.image_sources(node.inputs[0])
            if src is None:
                return None
            per = len(src) // node.attrs["num_outputs"]
            return src[node.attrs["index"] * per:(node.attrs["index"] + 1) * per]
        if node.op == "Concat" and node.attrs.get("dim", 1) == 0:
            parts = [self.image_sources(i) for i in node.inputs]
            return None if any(q is None for q in parts) else [x for q in parts for x in q]
        return None

    @staticmethod
    def _runs(src):
        """[(var, first index, count, position in the list)] for maximal runs of consecutive images of one variable"""
        runs, i = [], 0
        while i < len(src):
            j = i
            while j + 1 < len(src) and src[j + 1] == (src[j][0], src[j][1] + 1):
                j += 1
            runs.append((src[i][0], src[i][1], j - i + 1, i))
            i = j + 1
        return runs

    def _img_ref(self, var, i0, n):
        r = "%s:%d:3:4:%d:%d" % (var, i0 * 3 * self.H * self.W * 4, self.H, self.W)
        return r if n == 1 else r + ":%d" % n

    def image_nhwc4(self, node, bn=None):
        src = self.image_sources(node)
        key = ("img", tuple(src), bn.name if bn is not None else None)
        if key in self.val:
            return self.val[key]
        v = View(self.new_buf(3, self.H, self.W, N=len(src)), 3)
        for var, i0, n, pos in self._runs(src):
            dst = v.image(pos, n)
            args = {"src": self._img_ref(var, i0, n), "dst": dst, "H": self.H, "W": self.W}
            if bn is not None:
                args.update({"bn": bn.name, "eps": bn.attrs["eps"], "fixg": int(bn.attrs["fix_gamma"])})
            self.emit("prep_rgb", args, [], [dst], nbytes=(12 + 16) * self.H * self.W, n=n)
        self.val[key] = v
        for n_ in (node,):
            self.absorbed.add(id(n_))
        return v

    def input_view(self, node):
        """View holding the value of `node` as a conv/pool input."""
        if self.image_sources(node) is not None:
            return self.image_nhwc4(node)
        if node.op == "null":
            if node.name == "feat_key":
                N, C, H, W = self.shape(node)
                return self.pbuf_view(self.feat_src, C, H, W, N=N)
            raise NotImplementedError("variable %s used as activation" % node.name)
        if node.op == "_split_out":       # batch slice of an activation (accel_101.py:47-49)
            parent = self.input_view(node.inputs[0])
            per = parent.nimg // node.attrs["num_outputs"]
            return parent.image(node.attrs["index"] * per, per)
        if id(node) not in self.val:
            raise NotImplementedError("value of %s (%s) was not materialised before use" % (node.name, node.op))
        return self.val[id(node)]

    # ---- conv-like anchors ----------------------------------------------------------------
    def lower_anchor(self, A):
        a = A.attrs
        op = A.op
        x = A.inputs[0]
        mode = "conv"
        if op == "Deconvolution":
            if a["kernel"] != (4, 4) or a["stride"] != (2, 2) or a["num_group"] != 1 or a["pad"] not in ((0, 0), (1, 1)):
                raise NotImplementedError("Deconvolution %s: only 4x4/2 pad 0|1 (and the 32x32/16 score upsampler)" % A.name)
            mode = "deconv2x"
        if op == "Convolution" and a["num_group"] != 1:
            raise NotImplementedError("grouped Convolution %s" % A.name)
        # BatchNorm on the raw image (bn_data) folds into the image converter
        if x.op == "BatchNorm" and self.image_sources(x.inputs[0]) is not None:
            xin = self.image_nhwc4(x.inputs[0], bn=x)
            self.absorbed.add(id(x))
        elif op == "DeformableConvolution":
            xin = None
        else:
            xin = None if id(A) in self.split_fusion else self.input_view(x)      # (a fused Concat has no buffer: below)
        widx = 2 if op == "DeformableConvolution" else 1
        wname = A.inputs[widx].name
        bias = None if a["no_bias"] else A.inputs[widx + 1].name

        cur, bn, mul, res, act, slope = A, None, None, None, 0, 0.1
        if id(A) in self.split_fusion:      # (_plan_split_fusion) W_right * other + b + warp(W_left * feat)
            _, Bw, x = self.split_fusion[id(A)]
            c0, c1 = self.shape(Bw)[1], self.shape(Bw)[1] + self.shape(x)[1]
            wr = "%s[:,%d:%d]" % (wname, c0, c1)
            self.derived[wr] = ("cin_slice", wname, "%d:%d" % (c0, c1))
            wname, xin, res = wr, self.input_view(x), self.fusion_res[id(A)]
        need_crop = mode == "deconv2x" and a["pad"] == (0, 0)
        cropped = False
        chain = []
        opname = A.name
        # linear-linear fold: a bias-free, activation-free 4x4/2 Deconvolution whose only consumer is a 1x1
        # Convolution (Accel-18/34 `feat_upsampling` -> `fc6`, accel_18.py:196-209) is ONE 4x4/2 deconvolution with
        # the composed weight  W'[ci,co',ky,kx] = sum_c Wd[ci,c,ky,kx] * Wf[co',c]  -- 3x fewer flops, and the
        # 2048-channel intermediate is never written.  The composed weight is a derived parameter (fold_params()).
        if self.fold_linear and mode == "deconv2x" and a["no_bias"] and not need_crop and id(A) not in self.head_ids:
            cons = self.consumers(A)
            if len(cons) == 1 and cons[0].op == "Convolution" and cons[0].inputs[0] is A:
                f = cons[0].attrs
                if f["kernel"] == (1, 1) and f["stride"] == (1, 1) and f["pad"] == (0, 0) and f["num_group"] == 1:
                    F = cons[0]
                    derived = "%s*%s" % (wname, F.inputs[1].name)
                    self.derived[derived] = ("deconv4x4s2*conv1x1", wname, F.inputs[1].name)
                    wname = derived
                    bias = None if f["no_bias"] else F.inputs[2].name
                    opname = "%s*%s" % (A.name, F.name)
                    cur = F
                    chain.append(F)
        while True:
            if id(cur) in self.head_ids:
                break
            cons = self.consumers(cur)
            if len(cons) != 1:
                break
            n = cons[0]
            if need_crop and not cropped:
                if n.op == "Crop" and n.attrs["offset"] == (1, 1) and cur is A:
                    _, _, h, w = self.shape(n)
                    _, _, hi, wi = self.shape(x)
                    if h not in (2 * hi, 2 * hi - 1) or w not in (2 * wi, 2 * wi - 1):
                        raise NotImplementedError("Crop %s after 4x4/2 deconvolution must keep 2x the input (or one row / "
                                                  "column less)" % n.name)
                    cropped = True
                    cur = n
                    chain.append(n)
                    continue
                raise NotImplementedError("Deconvolution %s pad 0 must be followed by Crop(offset=(1,1))" % A.name)
            if n.op == "BatchNorm" and bn is None and mul is None and res is None and act == 0:
                bn = n
            elif n.op == "_mul_scalar" and res is None and act == 0:
                mul = (mul or 1.0) * n.attrs["scalar"]
            elif n.op in ("broadcast_add", "elemwise_add") and res is None and act == 0:
                other = n.inputs[1] if n.inputs[0] is cur else n.inputs[0]
                if other.op != "null" and id(other) not in self.val:
                    break
                res = self.input_view(other)
            elif n.op == "Activation" and n.attrs["act_type"] == "relu" and act == 0:
                act = 1
            elif n.op == "LeakyReLU" and n.attrs["act_type"] == "leaky" and act == 0:
                act, slope = 2, n.attrs["slope"]
            else:
                break
            cur = n
            chain.append(n)
        if need_crop and not cropped:
            raise NotImplementedError("Deconvolution %s pad 0 without Crop" % A.name)

        # key frame: the head's 1x1 conv on the propagated feature also leaves its raw linear image W*feat in the
        # derived buffer `featG`; non-key frames then warp that image instead of re-running the conv (lower_warp)
        feat_image = None
        if op == "Convolution" and x.op != "null" and id(x) in self.head_ids and x.name == "res5c_relu" \
                and self._head_on_feature(A, x) is not None and chain == [self._head_on_feature(A, x)] \
                and bn is None and mul is None and res is None:
            _, co_, ho_, wo_ = self.shape(A)
            feat_image = self._feat_image(A, co_, ho_, wo_)
        out = self.dest_for(cur)
        # dual output: relu(bn(.)) of the value for the next pre-activation unit
        out2, bn2, chain2 = None, None, []
        if feat_image is None:
            for n in self.consumers(cur):
                if n.op == "BatchNorm" and id(n) not in self.absorbed:
                    c2 = self.consumers(n)
                    if len(c2) == 1 and c2[0].op == "Activation" and c2[0].attrs["act_type"] == "relu" \
                            and id(n) not in self.head_ids:
                        bn2, r2 = n, c2[0]
                        out2 = self.dest_for(r2)
                        chain2 = [n, r2]
                        break

        _, cin, hi, wi = self.shape(x)
        nb, cout, ho, wo = self.shape(cur)
        args = {"name": opname, "out": out, "w": wname, "act": act, "slope": slope, "cin": cin, "cout": cout, "mode": mode}
        bias2 = None
        if feat_image is not None:
            out2, out, act, bias2, bias = out, feat_image, 0, bias, None
            args.update({"out": out, "act": 0})
        reads = [res]
        if op == "DeformableConvolution":
            off = A.inputs[1]
            xv, offv = self.input_view(x), self.input_view(off)
            kh, kw = a["kernel"]
            cp = _r4(cin)
            cols = View(self.new_buf(kh * kw * cp, ho, wo, N=nb), kh * kw * cp)
            self.emit("dcn_cols", {"name": A.name + "_cols", "in": xv, "off": offv, "out": cols,
                                   "k": "%d,%d" % a["kernel"], "s": "%d,%d" % a["stride"], "p": "%d,%d" % a["pad"],
                                   "d": "%d,%d" % a["dilate"], "dg": a["num_deformable_group"]},
                      [xv, offv], [cols], nbytes=4.0 * ho * wo * kh * kw * cp * 2, n=nb)
            args.update({"in": cols, "mode": "cols", "wk": "%d,%d" % a["kernel"], "k": "1,1"})
            reads.append(cols)
            flops = 2.0 * ho * wo * cout * cin * kh * kw
        elif mode == "deconv2x":
            args["in"] = xin
            reads.append(xin)
            flops = 2.0 * hi * wi * cout * cin * 16
        else:
            args.update({"in": xin, "k": "%d,%d" % a["kernel"], "s": "%d,%d" % a["stride"],
                         "p": "%d,%d" % a["pad"], "d": "%d,%d" % a["dilate"]})
            reads.append(xin)
            flops = 2.0 * ho * wo * cout * cin * a["kernel"][0] * a["kernel"][1]
        if id(A) in self.flow_ancestors:
            args["wb3"] = 0
        if bias:
            args["bias"] = bias
        if bn is not None:
            args.update({"bn": bn.name, "eps": bn.attrs["eps"], "fixg": int(bn.attrs["fix_gamma"])})
        if mul is not None:
            args["mul"] = mul
        if res is not None:
            args["res"] = res
        writes = [out]
        if bias2 is not None:
            args.update({"out2": out2, "bias2": bias2})
            writes.append(out2)
        elif out2 is not None:
            args.update({"out2": out2, "bn2": bn2.name, "eps2": bn2.attrs["eps"], "fixg2": int(bn2.attrs["fix_gamma"])})
            writes.append(out2)
        # algorithmic HBM bytes of the launch: every operand once (input, weights, residual, outputs)
        kk = a["kernel"][0] * a["kernel"][1]
        in_elems = ho * wo * kk * _r4(cin) if op == "DeformableConvolution" else hi * wi * cin
        if mode == "conv" and op == "Convolution":
            # a kernel smaller than its stride never reads the pixels between its taps (the stride-2 1x1 layers res3a / res4a
            # `branch1`, `branch2a`: a quarter of the input): only what is sampled counts as algorithmic traffic
            sy, sx = a["stride"]
            ky = (a["kernel"][0] - 1) * a["dilate"][0] + 1
            kx = (a["kernel"][1] - 1) * a["dilate"][1] + 1
            if ky < sy or kx < sx:
                in_elems = min(in_elems, ho * min(ky, sy) * wo * min(kx, sx) * cin)
        w_elems = cout * cin * (16 if mode == "deconv2x" else kk)
        nbytes = 4.0 * (in_elems + ho * wo * cout * (1 + (res is not None) + (out2 is not None)))
        args["_elems"] = {"in": in_elems, "out": ho * wo * cout, "w": w_elems, "n": nb}
        self.emit("conv", args, [r for r in reads if r is not None], writes, flops=flops, nbytes=nbytes, nbytes_fixed=4.0 * w_elems, n=nb)
        self.absorbed.add(id(A))
        if feat_image is not None:
            out, out2 = out2, None     # the graph value of the chain is the biased, activated copy
        for n in chain:
            self.absorbed.add(id(n))
            self.val[id(n)] = out
        self.val[id(A)] = out
        self.val[id(cur)] = out
        for n in chain2:
            self.absorbed.add(id(n))
        if chain2:
            self.val[id(chain2[-1])] = out2
        self.finish_value(cur, out)

    def finish_value(self, node, view):
        """post-processing common to every materialised value"""
        if id(node) in self.head_ids:
            self.outputs[node.name + "_output"] = view

    # ---- pooling ------------------------------------------------------------------------------
    def lower_pool(self, P):
        a = P.attrs
        x = P.inputs[0]
        if x.op == "Concat" and all(i.op == "_div_scalar" for i in x.inputs):
            return self.lower_flow_input(P, x)
        xin = self.input_view(x)
        cur, bn, relu, chain = P, None, 0, []
        while id(cur) not in self.head_ids:
            cons = self.consumers(cur)
            if len(cons) != 1:
                break
            n = cons[0]
            if n.op == "BatchNorm" and bn is None and not relu:
                bn = n
            elif n.op == "Activation" and n.attrs["act_type"] == "relu" and not relu and bn is not None:
                relu = 1
            else:
                break
            cur = n
            chain.append(n)
        if bn is not None and not relu:
            # BN without ReLU after a pool does not occur; keep the pool plain
            cur, bn, chain = P, None, []
        out = self.dest_for(cur)
        args = {"name": P.name, "in": xin, "out": out, "kind": a["pool_type"], "k": "%d,%d" % a["kernel"],
                "s": "%d,%d" % a["stride"], "p": "%d,%d" % a["pad"], "act": relu}
        if bn is not None:
            args.update({"bn": bn.name, "eps": bn.attrs["eps"], "fixg": int(bn.attrs["fix_gamma"])})
        nb, c, ho, wo = self.shape(P)
        _, _, hi, wi = self.shape(x)
        reads = [xin]
        self.emit("pool", args, reads, [out], nbytes=4.0 * c * (hi * wi + ho * wo), n=nb)
        for n in [P] + chain:
            self.absorbed.add(id(n))
            self.val[id(n)] = out
        self.finish_value(cur, out)

    def lower_flow_input(self, P, cat):
        a = P.attrs
        if a["pool_type"] != "avg" or a["kernel"] != (2, 2) or a["stride"] != (2, 2) or a["pad"] != (0, 0):
            raise NotImplementedError("FlowNet input pooling must be avg 2x2/2")
        srcs = []
        for d in cat.inputs:
            if abs(d.attrs["scalar"] - 255.0) > 0 or self.image_sources(d.inputs[0]) is None:
                raise NotImplementedError("FlowNet input must be image / 255.0")
            srcs.append(self.image_sources(d.inputs[0]))
            self.absorbed.add(id(d))
            self.absorbed.add(id(d.inputs[0]))
        self.absorbed.add(id(cat))
        cur, prev = srcs
        if len(cur) != len(prev):
            raise NotImplementedError("FlowNet input: %d current vs %d reference images" % (len(cur), len(prev)))
        nb = len(cur)
        out = View(self.new_buf(6, self.H // 2, self.W // 2, Cs=8, N=nb), 6)
        # one launch per run of image pairs that is contiguous in both sources (the test graphs: one launch)
        i = 0
        while i < nb:
            j = i
            while j + 1 < nb and cur[j + 1] == (cur[j][0], cur[j][1] + 1) and prev[j + 1] == (prev[j][0], prev[j][1] + 1):
                j += 1
            n = j - i + 1
            dst = out.image(i, n)
            self.emit("prep_flow", {"cur": self._img_ref(cur[i][0], cur[i][1], n), "prev": self._img_ref(prev[i][0], prev[i][1], n),
                                    "dst": dst, "H": self.H, "W": self.W}, [], [dst],
                      nbytes=24.0 * self.H * self.W + 8.0 * self.H * self.W, n=n)
            i = j + 1
        self.absorbed.add(id(P))
        self.val[id(P)] = out

    # ---- warp ---------------------------------------------------------------------------------------
    def lower_warp(self, B):
        """GridGenerator(warp)+BilinearSampler, or the registered `FlowWarp` custom op."""
        if B.op == "Custom":
            feat, flow_node = B.inputs
            grid = B
        else:
            feat, grid = B.inputs
            sl = None
            if grid.op == "_split_out":          # training graphs: one grid per intermediate frame (accel_18.py:53-57)
                sl, grid = grid, grid.inputs[0]
            if grid.op != "GridGenerator":
                raise NotImplementedError("BilinearSampler grid must come from GridGenerator(warp)")
            flow_node = grid.inputs[0]
        flow = self.input_view(flow_node)
        if B.op != "Custom" and sl is not None:
            per = flow.nimg // sl.attrs["num_outputs"]
            flow = flow.image(sl.attrs["index"] * per, per)
            self.absorbed.add(id(sl))
        fin = self.input_view(feat)
        nb, C, H, W = self.shape(B)
        pingpong = self.feat_dst is not None and id(B) in self.head_ids and feat.op == "null" and feat.name == "feat_key" \
            and id(B) not in self.concat_slot
        out = self.pbuf_view(self.feat_dst, C, H, W, N=nb) if pingpong else self.dest_for(B)
        if flow.nimg != fin.nimg:
            raise NotImplementedError("warp %s: %d feature images vs %d flow fields" % (B.name, fin.nimg, flow.nimg))
        self.emit("warp", {"name": B.name, "feat": fin, "flow": flow, "out": out}, [fin, flow], [out],
                  nbytes=2.0 * 4 * C * H * W + 8.0 * H * W, n=nb)
        self.absorbed.update((id(B), id(grid)))
        self.val[id(B)] = out
        if pingpong:
            self.outputs[B.name + "_output"] = out      # the propagated feature of the next frame, where the warp wrote it
        elif id(B) in self.head_ids:
            # the propagated feature: becomes `feat` for the next frame (demo.py:241-243)
            dst = self.pbuf_view("feat", C, H, W, N=nb)
            self.emit("copy", {"src": out, "dst": dst}, [out], [dst], nbytes=8.0 * C * H * W, n=nb)
            self.outputs[B.name + "_output"] = dst
        # Warp and a 1x1 convolution commute (both linear, the bilinear weights do not depend on the channel):
        #   relu(W * warp(F) + b) = relu(warp(W * F) + b),
        # out-of-image taps contribute zero on both sides, so the bias stays outside the warp.  The head's fc6 on the
        # warped feature (34.4 GFLOP at 1024x2048) becomes a warp of the 1024-channel image featG = W*F that the key
        # plan (or, after a host upload of `feat`, the init:featG plan) left in HBM; the warped image is also the next
        # frame's featG (iterative warping: F_t = warp(F_t-1)  =>  W*F_t = warp(W*F_t-1)).
        if id(B) in self.fusion_of_warp:
            # feature fusion over Concat(this warp, other) (_plan_split_fusion): warp the image featC = W_left * feat as well; the fusion
            # convolution takes it as its residual
            conv = self.fusion_of_warp[id(B)]
            wsrc = conv.inputs[1].name
            wl = "%s[:,0:%d]" % (wsrc, C)
            self.derived[wl] = ("cin_slice", wsrc, "0:%d" % C)
            _, Cg, _, _ = self.shape(conv)
            g_in = self._feat_image(conv, Cg, H, W, name="featC", wname=wl, cin=C)
            g_out = self.pbuf_view("featC" if self.feat_slot == 1 else "featC_b", Cg, H, W) if pingpong \
                else View(self.new_buf(Cg, H, W), Cg)
            self.emit("warp", {"name": B.name + "*" + conv.name, "feat": g_in, "flow": flow, "out": g_out}, [g_in, flow], [g_out],
                      nbytes=2.0 * 4 * Cg * H * W + 8.0 * H * W)
            if not pingpong:
                g_dst = self.pbuf_view("featC", Cg, H, W)
                self.emit("copy", {"src": g_out, "dst": g_dst}, [g_out], [g_dst], nbytes=8.0 * Cg * H * W)
            self.fusion_res[id(conv)] = g_out
        heads = [c for c in self.consumers(B) if self._head_on_feature(c, B) is not None]
        if feat.op == "null" and feat.name == "feat_key" and id(B) in self.head_ids \
                and len(self.consumers(B)) == 1 and len(heads) == 1:
            conv = heads[0]
            relu = self._head_on_feature(conv, B)
            _, Cg, _, _ = self.shape(conv)
            g_in = self._feat_image(conv, Cg, H, W)
            g_out = self.pbuf_view("featG" if self.feat_slot == 1 else "featG_b", Cg, H, W) if pingpong \
                else View(self.new_buf(Cg, H, W), Cg)
            act_out = self.dest_for(relu)
            self.emit("warp", {"name": B.name + "*" + conv.name, "feat": g_in, "flow": flow, "out": g_out, "out2": act_out,
                               "bias": conv.inputs[2].name}, [g_in, flow], [g_out, act_out],
                      nbytes=3.0 * 4 * Cg * H * W + 8.0 * H * W)
            if not pingpong:
                g_dst = self.pbuf_view("featG", Cg, H, W)
                self.emit("copy", {"src": g_out, "dst": g_dst}, [g_out], [g_dst], nbytes=8.0 * Cg * H * W)
            self.absorbed.update((id(conv), id(relu)))
            self.val[id(conv)] = act_out
            self.val[id(relu)] = act_out

    # ---- score tail --------------------------------------------------------------------------------
    @staticmethod
    def _is_upsampler(n):
        return n.op == "Deconvolution" and n.attrs["kernel"] == (32, 32) and n.attrs["stride"] == (16, 16) \
            and n.attrs["num_group"] == n.attrs["num_filter"] and n.attrs["pad"] == (0, 0) and n.attrs["no_bias"]

    def _tail_branch(self, crop, first=True):
        if crop.op != "Crop" or crop.attrs["offset"] != (8, 8) or not self._is_upsampler(crop.inputs[0]):
            return None
        up = crop.inputs[0]
        _, n, h, w = self.shape(crop)
        hs, ws = self.shape(up.inputs[0])[2:]
        # the first (left) map is exactly H/16 x W/16; the second may be one row / column larger: the stride-32 correction
        # branch upsampled 2x at frame sizes that are multiples of 16 but not of 32 (Crop(8, 8) to the frame drops the excess)
        ok = (hs, ws) == (self.H // 16, self.W // 16) if first else (self.H // 16 <= hs <= self.H // 16 + 1 and self.W // 16 <= ws <= self.W // 16 + 1)
        if (h, w) != (self.H, self.W) or self.H % 16 or self.W % 16 or not ok:
            raise NotImplementedError("score upsampling must map H/16 x W/16 scores onto the full image")
        return up

    def lower_tail(self, node, crops, corr=None, softmax=None):
        ups = [self._tail_branch(c, first=(i == 0)) for i, c in enumerate(crops)]
        if any(u is None for u in ups):
            raise NotImplementedError("unsupported score tail at %s" % node.name)
        ncls = ups[0].attrs["num_filter"]
        nb = self.shape(ups[0])[0]
        sfx = "" if nb == 1 else ":%d" % nb
        logits = self.pbuf_view("logits", ncls, self.H, self.W, Cs=4, N=nb)
        self.pbufs["logits"] = nb * ncls * self.H * self.W * 4
        labels = self.pbuf_view("labels", 1, self.H, self.W, Cs=4, N=nb)
        self.pbufs["labels"] = (nb * self.H * self.W + 255) // 256 * 256
        left = self.input_view(ups[0].inputs[0])
        args = {"name": node.name, "left": left, "wl": ups[0].inputs[1].name, "H": self.H, "W": self.W, "ncls": ncls,
                "logits": "logits:0:%d:4:%d:%d%s" % (ncls, self.H, self.W, sfx),
                "labels": "labels:0:1:4:%d:%d%s" % (self.H, self.W, sfx)}
        reads = [left]
        flops = 0.0
        if corr is not None:
            right = self.input_view(ups[1].inputs[0])
            args.update({"right": right, "wr": ups[1].inputs[1].name, "cw": corr.inputs[1].name, "cb": corr.inputs[2].name})
            reads.append(right)
            flops = 2.0 * ncls * 2 * ncls * self.H * self.W
        if softmax is not None:
            args["softmax"] = 1      # SoftmaxOutput(multi_output=True) at test time: softmax over the class axis
        self.emit("score_tail", args, reads, [], flops=flops, nbytes=4.0 * ncls * self.H * self.W + self.H * self.W, n=nb)
        for n in [node] + list(crops) + ups + ([corr] if corr is not None else []) + ([softmax] if softmax is not None else []):
            self.absorbed.add(id(n))
        self.outputs[(softmax or node).name + "_output"] = "logits"

    # ---- driver ---------------------------------------------------------------------------------------
    def run(self):
        self.plan_concats()
        for n in self.nodes:
            if id(n) in self.absorbed or n.op in ("null", "_group"):
                continue
            op = n.op
            if op == "Deconvolution" and self._is_upsampler(n):
                continue
            if op == "Crop" and self._is_upsampler(n.inputs[0]):
                cons = self.consumers(n)
                if id(n) in self.head_ids:
                    self.lower_tail(n, [n])
                elif len(cons) == 1 and cons[0].op == "SoftmaxOutput" and id(cons[0]) in self.head_ids \
                        and cons[0].attrs.get("multi_output"):
                    self.lower_tail(n, [n], softmax=cons[0])
                continue
            if op == "Concat":
                continue   # score concat: handled at the correction conv; batch concat of images: at its consumer
            if op == "_split_out":
                # a view of its input, resolved where it is consumed -- except as a member of a channel Concat, whose
                # buffer nobody else fills (accel_101.py:48-49,66: the current frame's half of the ResNet batch)
                if id(n) in self.concat_slot and self.image_sources(n) is None:
                    src, dst = self.input_view(n), self.concat_slot[id(n)]
                    nb, C, H, W = self.shape(n)
                    self.emit("copy", {"src": src, "dst": dst}, [src], [dst], nbytes=8.0 * C * H * W, n=nb)
                    self.val[id(n)] = dst
                continue
            if op == "SoftmaxOutput" and id(n) in self.absorbed:
                continue
            if op == "Convolution" and n.inputs[0].op == "Concat" and n.inputs[0].attrs.get("dim", 1) == 1 \
                    and id(n.inputs[0]) not in self.val and id(n) not in self.split_fusion:
                cat = n.inputs[0]
                if a_is_1x1(n) and len(cat.inputs) == 2:
                    cons = self.consumers(n)
                    sm = cons[0] if (len(cons) == 1 and cons[0].op == "SoftmaxOutput" and id(cons[0]) in self.head_ids
                                     and cons[0].attrs.get("multi_output") and id(n) not in self.head_ids) else None
                    self.lower_tail(n, cat.inputs, corr=n, softmax=sm)
                    self.absorbed.add(id(cat))
                    continue
                raise NotImplementedError("Convolution %s over an unsupported Concat" % n.name)
            if op in ("Convolution", "Deconvolution", "DeformableConvolution"):
                self.lower_anchor(n)
            elif op == "Pooling":
                self.lower_pool(n)
            elif op == "BilinearSampler":
                self.lower_warp(n)
            elif op == "Custom" and getattr(n.attrs["prop"], "lowering", None) == "warp":
                self.lower_warp(n)
            elif op == "Custom":
                raise NotImplementedError("Custom op %s (%s) runs on the host and cannot sit inside a device plan"
                                          % (n.name, n.attrs["op_type"]))
            elif op in ("_div_scalar", "GridGenerator"):
                continue   # absorbed by prep_flow / warp when their consumer is lowered
            elif op == "BatchNorm" and n.inputs[0].op == "null":
                continue   # bn_data: folded when its conv is lowered
            else:
                raise NotImplementedError("no lowering for %s (%s): not produced by a fusable anchor" % (n.name, op))
        for h in self.heads:
            if h.op == "null":
                self.outputs[h.name] = "input:" + h.name
        if self.store_f16:
            self.assign_storage()
        self.assign_offsets()
        return self

    # ---- half activation storage (f16-mode plans, BASELINE config 5) ------------------------------------------------------------
    @staticmethod
    def half_capable(args):
        """A convolution that can take half inputs / residuals and write half outputs: an f16-mode layer (Cin % 8 == 0 and more
        than 4 output channels -- accel_hip.cpp finalize_conv) that conv_b3d.hip can run (Cin % 16 == 0: its K stage of 16 must
        lie inside one tap), with a single output (the dual-output epilogue of the pre-activation nets stays fp32)."""
        cin = _r4(int(args["cin"]))           # (mode=cols: the column buffer holds kh*kw taps of cin channels each, a 1x1 GEMM over them)
        return cin % 16 == 0 and _r4(int(args["cout"])) > 4 and "out2" not in args and int(args.get("tile", -1)) < 0

    def assign_storage(self):
        """Specification of the half-storage mode (restated by oracle/graphs.py STORE_F16): an arena buffer is kept as HALF iff every
        op that writes it is a half-capable convolution writing its (single) output there (or the deformable sampler writing the
        column buffer of a deformable layer) and every op that reads it is a half-capable convolution reading it as data input or as
        residual.  Such a buffer's values are rounded to half (RTNE)
        when they are stored, after the layer's whole epilogue (scale / shift, residual, activation); every reader sees the
        rounded value.  For a consumer convolution that is what its loader would have rounded the fp32 value to anyway; the
        residual reader is the one place where the function changes against fp32 storage.  Everything else -- persistent
        buffers (images, `feat`, logits), concat buffers with ragged channel counts, inputs of pools / warps / the deformable
        sampler / the narrow flow predictors -- stays fp32."""
        uses = {}
        for kind, args in self.ops:
            for k, v in args.items():
                if isinstance(v, View) and v.buf.space == "A":
                    uses.setdefault(id(v.buf), [v.buf, []])[1].append((kind, k, args, v))
        self.half_bufs = []
        for buf, us in uses.values():
            ok = buf.Cs % 8 == 0
            for kind, k, args, v in us:
                if kind == "dcn_cols" and k == "out":
                    continue        # the column buffer of a deformable layer: its only reader (checked like any reader) is the GEMM,
                                    # which rounds the sampled columns to half in any case -- storing them rounded changes no value
                ok = ok and kind == "conv" and k in ("in", "out", "res") and self.half_capable(args) and v.coff % 8 == 0
            if ok and any(k == "out" for _, k, _, _ in us):
                buf.esize = 2
                self.half_bufs.append(buf)
        # algorithmic bytes of the convolutions that touch half buffers: every operand once, at its stored width
        for kind, args in self.ops:
            if kind == "dcn_cols" and args["out"].buf.esize == 2 and "bytes" in args:
                args["bytes"] = "%.6g" % (0.75 * float(args["bytes"]))      # gathered reads at 4 bytes + the column store at 2
            if kind != "conv" or "_elems" not in args:
                continue
            e = args["_elems"]
            es = lambda key: (args[key].buf.esize if isinstance(args.get(key), View) else 4)
            nbytes = e["n"] * (e["in"] * es("in") + e["out"] * es("out") + (e["out"] * es("res") if "res" in args else 0)
                               + (e["out"] * 4 if "out2" in args else 0)) + 4.0 * e["w"]
            args["bytes"] = "%.6g" % nbytes

    def assign_offsets(self):
        """Greedy first-fit arena packing over [first, last] op-index lifetimes (list order = execution order)."""
        live = [b for b in self.bufs if b.first is not None]
        import os
        if os.environ.get("ACCEL_ARENA_NO_REUSE") == "1":
            # diagnostics: every buffer keeps private space, so the arena after a run holds every op's own output
            total = 0
            for b in live:
                b.off = total
                total += (b.nbytes + ALIGN - 1) // ALIGN * ALIGN
            self.arena_bytes = total
            return

        def pack(bufs, base):
            placed, total = [], 0
            for b in sorted(bufs, key=lambda b: -b.nbytes):
                size = (b.nbytes + ALIGN - 1) // ALIGN * ALIGN
                taken = sorted((q.off - base, q.off - base + (q.nbytes + ALIGN - 1) // ALIGN * ALIGN) for q in placed
                               if not (q.last < b.first or b.last < q.first))
                off = 0
                for s0, e0 in taken:
                    if off + size <= s0:
                        break
                    off = max(off, e0)
                b.off = base + off
                placed.append(b)
                total = max(total, off + size)
            return total

        self.arena_bytes = pack(live, 0)

    def text(self, graph=True, conv_dtype="f32"):
        lines = ["# accel_amd plan: %d ops, arena %.1f MB, %.2f GFLOP" % (len(self.ops), self.arena_bytes / 1e6, self.total_flops / 1e9)]
        if not graph:
            lines.append("option graph=0")
        if conv_dtype == "f16":
            lines.append("option dtype=f16")   # convolutions on the fp16 matrix cores (fp32 storage + accumulate)
        elif conv_dtype == "bf16x3":
            lines.append("option dtype=bf16x3")   # fp32 operands as three bf16 terms on the bf16 matrix cores (fp32-equivalent)
        elif conv_dtype != "f32":
            raise ValueError("conv_dtype must be 'f32', 'f16' or 'bf16x3'")
        lines.append("meta feat_c=2048 feat_h=%d feat_w=%d feat_n=%d" % (self.H // 16, self.W // 16, self.N))
        lines.append("arena bytes=%d" % max(self.arena_bytes, ALIGN))
        for name, nbytes in sorted(self.pbufs.items()):
            src = self.derived_bufs.get(name, {}).get("from")
            lines.append("pbuf name=%s bytes=%d%s" % (name, nbytes, " from=%s" % src if src else ""))
        # Range slots (csrc/range.h, kernels.h ConvParams::xr / yr): every physical buffer gets an id; a convolution names the buffer it
        # reads (`xr=`), every op the buffers it writes (`yr=`, `y2r=`).  The library gives a slot to the buffers an fp16x2-form convolution
        # reads and has their writers raise it to the largest |value| they store -- in the run that uses it.
        pb_ids = {}

        def rid(view):
            if not isinstance(view, View):
                return None
            b = view.buf
            if b.space == "A":
                return b.id
            return 1000000 + pb_ids.setdefault(b.space, len(pb_ids))
        range_keys = {"conv": (("in", "xr"), ("out", "yr"), ("out2", "y2r")), "pool": (("out", "yr"),), "dcn_cols": (("out", "yr"),),
                      "warp": (("out", "yr"), ("out2", "y2r")), "prep_rgb": (("dst", "yr"),), "prep_flow": (("dst", "yr"),),
                      "copy": (("dst", "yr"),)}
        for kind, args in self.ops:
            toks = [kind]
            for src, key in range_keys.get(kind, ()):
                r = rid(args.get(src))
                if r is not None:
                    toks.append("%s=%d" % (key, r))
            for k, v in args.items():
                if k.startswith("_"):
                    continue                    # lowering-internal bookkeeping
                if isinstance(v, View):
                    v = v.ref()
                elif isinstance(v, float):
                    v = repr(v)
                toks.append("%s=%s" % (k, v))
            lines.append(" ".join(toks))
        return "\n".join(lines) + "\n"


def a_is_1x1(n):
    a = n.attrs
    return a["kernel"] == (1, 1) and a["stride"] == (1, 1) and a["pad"] == (0, 0) and not a["no_bias"]


def fold_params(derived, params):
    """Composed weights of the linear-linear folds a Lowering recorded (`Lowering.derived`), computed in float64
    on the host at bind time -- weight preparation, like the BatchNorm fold.  `params`: name -> array."""
    import numpy as np
    out = {}
    for name, (kind, wd_name, wf_name) in derived.items():
        if kind == "cin_slice":      # input channels [lo, hi) of a convolution weight (the two halves of a fusion over a Concat)
            w = params[wd_name]
            w = np.asarray(w.asnumpy() if hasattr(w, "asnumpy") else w, dtype=np.float32)
            lo, hi = (int(v) for v in wf_name.split(":"))
            out[name] = np.ascontiguousarray(w[:, lo:hi])
            continue
        if kind != "deconv4x4s2*conv1x1":
            raise NotImplementedError(kind)
        wd, wf = params[wd_name], params[wf_name]
        wd = np.asarray(wd.asnumpy() if hasattr(wd, "asnumpy") else wd, dtype=np.float64)   # (Cin, C, 4, 4)
        wf = np.asarray(wf.asnumpy() if hasattr(wf, "asnumpy") else wf, dtype=np.float64)   # (Cout, C, 1, 1)
        cin, c, kh, kw = wd.shape
        w = wd.transpose(0, 2, 3, 1).reshape(cin * kh * kw, c) @ wf.reshape(wf.shape[0], c).T
        out[name] = np.ascontiguousarray(w.reshape(cin, kh, kw, -1).transpose(0, 3, 1, 2)).astype(np.float32)
    return out


def init_plan_text(name, d):
    """Plan of role `init:<name>` that rebuilds a derived persistent buffer from its source (run by
    accel_plan_run when a plan reads the buffer while it is stale, e.g. after a host upload of `feat`)."""
    H, W, cin, cout, N = d["H"], d["W"], d["cin"], d["cout"], d.get("N", 1)
    sfx = "" if N == 1 else ":%d" % N
    return "\n".join([
        "# accel_amd plan: rebuild %s = %s * %s" % (name, d["w"], d["from"]),
        "option graph=0",
        "arena bytes=%d" % ALIGN,
        "pbuf name=%s bytes=%d" % (d["from"], N * H * W * _r4(cin) * 4),
        "pbuf name=%s bytes=%d from=%s" % (name, N * H * W * _r4(cout) * 4, d["from"]),
        "conv name=init_%s out=%s:0:%d:%d:%d:%d%s w=%s act=0 slope=0.1 cin=%d cout=%d mode=conv in=%s:0:%d:%d:%d:%d%s "
        "k=1,1 s=1,1 p=0,0 d=1,1 flops=%g" % (name, name, cout, _r4(cout), H, W, sfx, d["w"], cin, cout,
                                           d["from"], cin, _r4(cin), H, W, sfx, 2.0 * N * H * W * cin * cout)]) + "\n"


def lower(sym, input_shapes, graph=True, conv_dtype="f32", fold_linear=True, feat_slot=None, store_f16=None):
    """store_f16 (f16-mode plans only; default: on unless ACCEL_F16_STORAGE=0): activations between half-capable convolutions are
    stored as half (Lowering.assign_storage)."""
    import os
    if store_f16 is None:
        store_f16 = os.environ.get("ACCEL_F16_STORAGE", "1") != "0"
    # (an f16-mode plan keeps the fusion of Accel-101 as the reference's one layer: the mode's specification rounds the operands of THAT
    # contraction -- oracle.graphs ROUND_F16 -- and the rebuild plan of a derived buffer runs in the default arithmetic)
    lw = Lowering(sym, input_shapes, fold_linear=fold_linear, feat_slot=feat_slot, store_f16=bool(store_f16) and conv_dtype == "f16",
                  fold_fusion=conv_dtype != "f16").run()
    return lw.text(graph=graph, conv_dtype=conv_dtype), lw
