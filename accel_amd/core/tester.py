"""Predictor / im_segment with the reference's signatures
(dff_deeplab/core/tester.py:22-35,158-171), running lowered plans on the HIP
executor instead of an MXNet MutableModule.

Contract kept from the reference:
  * Predictor(symbol, data_names, label_names, context, max_data_shapes,
    provide_data, provide_label, arg_params, aux_params): bind + init_params;
    missing / mis-shaped parameters raise at construction (allow_missing=False).
  * predict(data_batch) -> [ {output_name: array} ] (one dict per device);
    inputs are copied into executor-owned buffers, outputs stay valid until the
    next forward; arrays are device handles whose .asnumpy() synchronises.
  * a key predictor and a cur predictor built on the same context with the SAME
    parameters share one model, hence the propagated feature in HBM: feeding
    `feat` back as `feat_key` (demo.py:241-243) costs nothing when it is the
    handle the previous predict returned.  The handle carries the buffer's write
    generation: if anything has overwritten the buffer since (another forward,
    another predictor pair), the handle's host copy is uploaded when it has one
    and the call fails loudly otherwise -- stale bytes are never read.  A host
    array is always uploaded.
Shapes are static per bind; a new (H, W) re-lowers and re-binds like
MutableModule.forward does on a shape change (module.py:1026-1042).
"""
import os

import numpy as np

from .. import lower as _lower
from .. import runtime
from ..mx.ndarray import DeviceArray

_MODELS = {}   # (device id, N, H, W, parameter token, owner) -> runtime.Model: the key and cur plans of one demo share buffers


def _device_id(context):
    ctx = context[0] if isinstance(context, (list, tuple)) else context
    return int(getattr(ctx, "device_id", 0) or 0)


def params_token(*dicts):
    """Content hash of parameter dicts (names, shapes, bytes).  Two predictors share a model -- and with it the
    repacked weights in HBM -- only when their parameters are the same BYTES; another network or another checkpoint
    gets its own model instead of overwriting same-named weights (fc6_weight, score_weight ... exist in every Accel
    variant).  ~0.1 s for the 0.45 GB of Accel-18."""
    try:
        import xxhash
        h = xxhash.xxh3_64()
    except ImportError:          # optional dependency: the standard library's blake2b is ~5x slower and just as good
        import hashlib
        h = hashlib.blake2b(digest_size=16)
    for d in dicts:
        for k in sorted(d):
            v = d[k]
            a = np.ascontiguousarray(v.asnumpy() if hasattr(v, "asnumpy") else v)
            h.update(("%s|%s|%s|" % (k, a.shape, a.dtype)).encode())
            h.update(memoryview(a).cast("B"))
    return h.hexdigest()


def shared_model(device_id, hw=None, token="", owner=None):
    """One model per device, frame size, parameter content and owner: persistent buffers (`data`, `feat`, `logits`...)
    are sized at the first bind and baked into captured graphs, so another resolution gets its own model.  `owner`
    scopes the sharing: a ClipRunner passes its own tag so that its key and cur predictor share buffers with each
    other and with nobody else; predictors built without one (reference-style code) share per device."""
    key = (device_id,) + tuple(hw or ()) + (token, owner)
    if key not in _MODELS:
        _MODELS[key] = runtime.Model(runtime.Context(device_id))
    return _MODELS[key]


def release_models(owner=None):
    """Frees the models (weights, arenas, persistent buffers) bound so far -- all of them, or one owner's."""
    for key in [k for k in _MODELS if owner is None or k[-1] == owner]:
        m = _MODELS.pop(key)
        m.close()
        m.ctx.close()


class Predictor(object):
    def __init__(self, symbol, data_names, label_names, context=None, max_data_shapes=None,
                 provide_data=None, provide_label=None, arg_params=None, aux_params=None, model=None, owner=None):
        self._symbol = symbol
        self._data_names = list(data_names)
        self.output_names = symbol.list_outputs()
        self._explicit_model = model
        self._owner = owner
        self._device_id = _device_id(context)
        self._model = model
        self._arg_params = arg_params or {}
        self._aux_params = aux_params or {}
        self._plans = {}      # (N, H, W) -> (runtime.Plan, Lowering, runtime.Model)
        self._variants = {}   # (N, H, W) -> ((plan, lowering) reading `feat`, (plan, lowering) reading `feat_b`)
        self._is_key = "feat_key" not in symbol.list_arguments() or "feat_key" in self.output_names \
            or any(n.startswith("res5c_relu") for n in self.output_names)
        shapes = dict(provide_data[0]) if provide_data else {}
        if max_data_shapes:
            for k, v in max_data_shapes[0]:
                shapes.setdefault(k, v)
        if provide_label and provide_label[0]:
            for k, v in provide_label[0]:
                shapes.setdefault(k, v)
        # inputs beyond data / data_key / feat_key (the training graphs: data_ref, eq_flag, label) bind at the given shapes
        self._extra_shapes = {k: tuple(v) for k, v in shapes.items()
                              if k not in ("data", "data_key", "feat_key") and k in symbol.list_arguments()}
        self._is_train = "data_ref" in symbol.list_arguments()
        if "data" in shapes:
            self._bind(tuple(shapes["data"])[2:], tuple(shapes["data"])[0])

    # -- bind = lower + finalize --------------------------------------------------------------
    def _check_params(self, lw_sym, input_shapes):
        arg_shapes, _, aux_shapes = lw_sym.infer_shape(**input_shapes)
        for name, shp in zip(lw_sym.list_arguments(), arg_shapes):
            if name in input_shapes or name.endswith("_label"):
                continue
            if name not in self._arg_params:
                raise RuntimeError("%s not initialized" % name)
            got = tuple(self._arg_params[name].shape)
            if got != tuple(shp):
                raise RuntimeError("shape inconsistent for %s inferred %s provided %s" % (name, shp, got))
        for name, shp in zip(lw_sym.list_auxiliary_states(), aux_shapes):
            if name not in self._aux_params:
                raise RuntimeError("%s not initialized" % name)
            got = tuple(self._aux_params[name].shape)
            if got != tuple(shp):
                raise RuntimeError("shape inconsistent for %s inferred %s provided %s" % (name, shp, got))

    def _bind(self, hw, N=1):
        """Static bind for (batch, H, W).  A batch > 1 stacks independent frames (one per clip): the convolutions run
        with M = N*Ho*Wo, which is what fills the chip on the stride-16 layers."""
        H, W, N = int(hw[0]), int(hw[1]), int(N)
        if (N, H, W) in self._plans:
            return self._plans[(N, H, W)]
        if H % 16 or W % 16:
            # the task head upsamples H/16 x W/16 scores by exactly 16 (Deconvolution 32x32/16 + Crop(8, 8)): the reference's own
            # shape inference fails on anything else; it only ever sees 1024x2048
            raise ValueError("image size %dx%d: frame sizes must be multiples of 16" % (H, W))
        feat_shape = (N, 2048, 1, 1) if self._is_key else (N, 2048, H // 16, W // 16)
        shapes = {"data": (N, 3, H, W), "data_key": (N, 3, H, W), "feat_key": feat_shape}
        shapes = {k: v for k, v in shapes.items() if k in self._symbol.list_arguments()}
        shapes.update(self._extra_shapes)
        self._check_params(self._symbol, shapes)
        model = self._explicit_model
        if getattr(self, "_token", None) is None:      # one hash per predictor, not per bound shape
            self._token = params_token(self._arg_params, self._aux_params)
        token = self._token
        if model is None:
            model = shared_model(self._device_id, (N, H, W), token, self._owner)
        if getattr(model, "_params_token", None) != token:
            if getattr(model, "_params_token", None) is not None and model.plans:
                raise runtime.AccelError("this model already holds plans bound to other parameters; bind another network "
                                         "or checkpoint on its own runtime.Model")
            model.set_params(self._arg_params, self._aux_params)
            model._params_token = token
        self._model = model
        # non-key graphs bind as a PAIR of plans that ping-pong the propagated feature between two buffer pairs (no
        # copy-back after the warp, lower.Lowering.__init__)
        pingpong = not self._is_key and not self._is_train
        # every plan is lowered for ONE stream (a two-stream lowering existed until round 3 and was removed: DESIGN.md 7)
        fold = os.environ.get("ACCEL_FOLD_LINEAR", "1") != "0"
        kw = dict(conv_dtype=os.environ.get("ACCEL_CONV_DTYPE", "f32"), fold_linear=fold)
        text, lw = _lower.lower(self._symbol, shapes, feat_slot=0 if pingpong else None, **kw)
        pingpong = pingpong and any(getattr(getattr(v, "buf", None), "space", None) == "feat_b" for v in lw.outputs.values())
        if not pingpong and lw.feat_slot is not None:
            text, lw = _lower.lower(self._symbol, shapes, **kw)
        if lw.derived:
            done = getattr(model, "_derived_from", {})
            todo = {k: v for k, v in lw.derived.items() if done.get(k) != token}
            for name, w in _lower.fold_params(todo, self._arg_params).items():
                model.set_param(name, w)
                done[name] = token
            model._derived_from = done
        for name, d in lw.derived_bufs.items():      # rebuild plans of the derived persistent buffers (featG = fc6_weight * feat; Accel-101: featC = corr_weight[:, :2048] * feat)
            if not self._is_key and ("init:" + name) not in model.plans:
                model.add_plan("init:" + name, _lower.init_plan_text(name, d)).finalize()
        role = "train" if self._is_train else "key" if self._is_key else "cur"      # key / cur: what accel_key_forward / accel_cur_forward look up
        if role in model.plans:
            role = "%s_%x" % (role, id(self) & 0xFFFFFF)
        plan = model.add_plan(role, text)
        plan.finalize()
        if pingpong:
            text_b, lw_b = _lower.lower(self._symbol, shapes, feat_slot=1, **kw)
            plan_b = model.add_plan(role + "_b", text_b)      # `cur_b`: what accel_cur_forward runs on odd non-key frames
            plan_b.finalize()
            self._variants[(N, H, W)] = ((plan, lw), (plan_b, lw_b))
        self._plans[(N, H, W)] = (plan, lw, model)
        return self._plans[(N, H, W)]

    # -- forward ---------------------------------------------------------------------------------
    def predict(self, data_batch):
        arrays = dict(zip(self._data_names, data_batch.data[0]))
        data = arrays["data"]
        N, _, H, W = tuple(data.shape)
        plan, lw, m = self._bind((H, W), N)
        self._model = m
        # Host->HBM copies are the dominant cost of the reference's per-frame loop (25 MB fp32 per image).  The key
        # graph does not read `data_key`, and on non-key frames `data_key` is the previous call's `data` array
        # (demo.py:176-181 builds it that way).  Arrays are immutable and carry a uid, so "the bytes of this input are
        # already in that HBM buffer" is an identity check: the image is then copied inside HBM (or not at all)
        # instead of crossing PCIe again.  An input announced with prefetch() is taken from its shadow buffer.
        res = m.__dict__.setdefault("_resident", {})
        pre = m.__dict__.setdefault("_prefetched", {})
        if not self._is_key:
            tag = _uid(arrays["data_key"])
            if tag is None or res.get("data_key") != tag:
                if tag is not None and res.get("data") == tag:
                    ptr, _ = m.buffer("data_key")
                    m.read_device("data", ptr, N * 3 * H * W * 4)
                else:
                    m.write("data_key", _host(arrays["data_key"]))
                res["data_key"] = tag
        tag = _uid(arrays["data"])
        if tag is None or res.get("data") != tag:
            if tag is not None and pre.get("data") == tag:
                m.commit("data")
            else:
                m.write("data", _host(arrays["data"]))
            res["data"] = tag
        pre.pop("data", None)
        for name in self._data_names:          # further image inputs (data_ref of the training graphs)
            if name not in ("data", "data_key", "feat_key") and name in lw.image_vars:
                m.write(name, _host(arrays[name]))
        if not self._is_key:
            fk = arrays["feat_key"]
            ref = getattr(fk, "device_ref", None)
            variants = self._variants.get((N, H, W))
            if ref and ref[0] is m and ref[1] in ("feat", "feat_b") and ref[2] == m.generation(ref[1]) \
                    and (ref[1] == "feat" or variants):
                # the handle IS the current content of one of the model's feature buffers: run the plan that reads it
                if variants:
                    plan, lw = variants[1 if ref[1] == "feat_b" else 0]
            elif ref and not getattr(fk, "has_host_copy", True):
                if ref[1] in ("feat", "feat_b") and ref[0] is not m and ref[2] == ref[0].generation(ref[1]) \
                        and ref[0].ctx.device_id == m.ctx.device_id:
                    # a handle of ANOTHER model on this GPU that is still current there: copy HBM to HBM
                    ref[0].ctx.sync()
                    src, n = ref[0].buffer(ref[1])
                    m.write_device("feat", src, N * 2048 * (H // 16) * (W // 16) * 4)
                else:
                    raise runtime.AccelError(
                        "feat_key is a device handle whose buffer has been overwritten since it was produced (outputs are "
                        "valid until the next forward that writes them); call .asnumpy() on it before that forward to "
                        "keep a copy, or give each predictor pair its own runtime.Model")
            else:
                self._upload_feat(_host(fk), N, H, W)
        plan.run()
        out = {}
        for name in self.output_names:
            d = lw.outputs.get(name)
            if isinstance(d, str) and d.startswith("input:"):
                out[name] = arrays[d[6:]]
            elif d == "logits":
                out[name] = self._logits_handle(N, H, W)
            else:
                out[name] = self._feat_handle(N, H, W, getattr(getattr(d, "buf", None), "space", "feat"))
        return [out]

    def _logits_handle(self, N, H, W, ncls=19):
        m = self._model

        def labels():
            # mx.ndarray.argmax returns float indices that the harness casts to uint8 at once (demo.py:245-246); the
            # fused kernel wrote uint8 labels, and they are handed over as they are (no 4x wider host copy per frame)
            return DeviceArray(shape=(N, H, W), fetch=lambda: m.read("labels", (N, H, W), np.uint8),
                               device_ref=(m, "labels", m.generation("labels")))
        return DeviceArray(shape=(N, ncls, H, W), fetch=lambda: m.read("logits", (N, ncls, H, W)),
                           device_ref=(m, "logits", m.generation("logits")), labels_of=labels)

    def _feat_handle(self, N, H, W, buf="feat"):
        m = self._model
        h, w = H // 16, W // 16

        def fetch():
            nhwc = m.read(buf, (N, h, w, 2048))
            return np.ascontiguousarray(nhwc.transpose(0, 3, 1, 2))
        gen = m.generation(buf)

        def fetch_checked():
            if m.generation(buf) != gen:
                raise runtime.AccelError("this feature handle is stale: its buffer has been written since (fetch it with "
                                         ".asnumpy() before the next forward that propagates a feature)")
            return fetch()
        return DeviceArray(shape=(N, 2048, h, w), fetch=fetch_checked, device_ref=(m, buf, gen))

    def _upload_feat(self, feat_nchw, N, H, W):
        f = np.asarray(feat_nchw, np.float32)
        if f.shape != (N, 2048, H // 16, W // 16):
            raise ValueError("feat_key shape %s does not match the bound graph" % (f.shape,))
        self._model.write("feat", np.ascontiguousarray(f.transpose(0, 2, 3, 1)))

    def prefetch(self, data_array):
        """Announce the NEXT call's `data` input: a page-locked array (mx.nd.array(.., ctx=mx.cpu_pinned())) starts
        crossing PCIe on the copy stream now, beside the running forward; the next predict() that receives this very
        array takes it from the shadow buffer.  Anything else is ignored (uploaded at predict time as usual)."""
        pb = getattr(data_array, "pinned", None)
        if pb is None or self._model is None:
            return False
        self._model.prefetch("data", pb)
        self._model.__dict__.setdefault("_prefetched", {})["data"] = data_array.uid
        return True

    def plan_for(self, H, W, N=1, slot=0):
        """(plan, lowering) bound for N frames of H x W; slot 1 = the ping-pong variant of a non-key graph that reads
        `feat_b` and writes `feat` (the one to run on every second non-key frame)."""
        plan, lw, _ = self._bind((H, W), N)
        if slot and (N, H, W) in self._variants:
            return self._variants[(N, H, W)][1]
        return plan, lw


def _uid(a):
    """content identity of an input array: DeviceArrays are immutable, raw numpy inputs have none (always uploaded)"""
    return getattr(a, "uid", None)


def _host(a):
    if hasattr(a, "asnumpy"):
        a = a.asnumpy()
    return np.ascontiguousarray(a, dtype=np.float32)


def im_segment(predictor, data_batch):
    """tester.py:158-171: returns (output_all, feat) where feat is the feature to
    propagate: res5c_relu_output after a key frame, warping_feat_output otherwise."""
    output_all = predictor.predict(data_batch)
    if 'res5c_relu_output' in output_all[0]:
        feat = output_all[0]['res5c_relu_output']
    elif 'warping_feat_output' in output_all[0]:
        feat = output_all[0]['warping_feat_output']
    else:
        feat = None
    return output_all, feat
