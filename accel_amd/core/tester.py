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
  * a key predictor and a cur predictor built on the same context share the
    propagated feature in HBM: feeding `feat` back as `feat_key`
    (demo.py:241-243) costs nothing when it is the handle the previous predict
    returned; a host array is uploaded instead.
Shapes are static per bind; a new (H, W) re-lowers and re-binds like
MutableModule.forward does on a shape change (module.py:1026-1042).
"""
import os

import numpy as np

from .. import lower as _lower
from .. import runtime
from ..mx.ndarray import DeviceArray

_MODELS = {}   # (device id, H, W) -> shared runtime.Model: the key and cur plans of one demo share buffers


def _device_id(context):
    ctx = context[0] if isinstance(context, (list, tuple)) else context
    return int(getattr(ctx, "device_id", 0) or 0)


def shared_model(device_id, hw=None):
    """One model per device and frame size: persistent buffers (`data`, `feat`, `logits`...) are sized
    at the first bind and baked into captured graphs, so another resolution gets its own model."""
    key = (device_id,) + tuple(hw or ())
    if key not in _MODELS:
        _MODELS[key] = runtime.Model(runtime.Context(device_id))
    return _MODELS[key]


def release_models():
    for m in _MODELS.values():
        m.close()
        m.ctx.close()
    _MODELS.clear()


class Predictor(object):
    def __init__(self, symbol, data_names, label_names, context=None, max_data_shapes=None,
                 provide_data=None, provide_label=None, arg_params=None, aux_params=None, model=None):
        self._symbol = symbol
        self._data_names = list(data_names)
        self.output_names = symbol.list_outputs()
        self._explicit_model = model
        self._device_id = _device_id(context)
        self._model = model
        self._arg_params = arg_params or {}
        self._aux_params = aux_params or {}
        self._plans = {}      # (H, W) -> (role, runtime.Plan, Lowering)
        self._is_key = "feat_key" not in symbol.list_arguments() or "feat_key" in self.output_names \
            or any(n.startswith("res5c_relu") for n in self.output_names)
        shapes = dict(provide_data[0]) if provide_data else {}
        if max_data_shapes:
            for k, v in max_data_shapes[0]:
                shapes.setdefault(k, v)
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
        if H % 128 or W % 128:
            raise ValueError("image size %dx%d: the FlowNet encoder/decoder needs multiples of 128" % (H, W))
        feat_shape = (N, 2048, 1, 1) if self._is_key else (N, 2048, H // 16, W // 16)
        shapes = {"data": (N, 3, H, W), "data_key": (N, 3, H, W), "feat_key": feat_shape}
        shapes = {k: v for k, v in shapes.items() if k in self._symbol.list_arguments()}
        self._check_params(self._symbol, shapes)
        model = self._explicit_model
        if model is None:
            model = shared_model(self._device_id, (N, H, W))
        loaded = getattr(model, "_loaded_from", set())
        if id(self._arg_params) not in loaded:
            model.set_params(self._arg_params, self._aux_params)
            loaded.add(id(self._arg_params))
            model._loaded_from = loaded
        self._model = model
        text, lw = _lower.lower(self._symbol, shapes, multi_stream=os.environ.get("ACCEL_MULTI_STREAM", "1") != "0",
                                conv_dtype=os.environ.get("ACCEL_CONV_DTYPE", "f32"),
                                fold_linear=os.environ.get("ACCEL_FOLD_LINEAR", "1") != "0")
        if lw.derived:
            done = getattr(model, "_derived_from", {})
            todo = {k: v for k, v in lw.derived.items() if done.get(k) != id(self._arg_params)}
            for name, w in _lower.fold_params(todo, self._arg_params).items():
                model.set_param(name, w)
                done[name] = id(self._arg_params)
            model._derived_from = done
        for name, d in lw.derived_bufs.items():      # rebuild plans of the derived persistent buffers (featG = fc6_weight * feat)
            if not self._is_key and ("init:" + name) not in model.plans:
                model.add_plan("init:" + name, _lower.init_plan_text(name, d)).finalize()
        role = "key" if self._is_key else "cur"      # the roles accel_key_forward / accel_cur_forward look up
        if role in model.plans:
            role = "%s_%x" % (role, id(self) & 0xFFFFFF)
        plan = model.add_plan(role, text)
        plan.finalize()
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
        # (demo.py:176-181 builds it that way): when it is the same object with the same content fingerprint the
        # image is copied inside HBM instead of crossing PCIe again.
        res = m.__dict__.setdefault("_resident", {})
        if not self._is_key:
            tag = _fingerprint(arrays["data_key"])
            if res.get("data_key") != tag:
                if res.get("data") == tag:
                    ptr, _ = m.buffer("data_key")
                    m.read_device("data", ptr, N * 3 * H * W * 4)
                else:
                    m.write("data_key", _host(arrays["data_key"]))
                res["data_key"] = tag
        tag = _fingerprint(arrays["data"])
        if res.get("data") != tag:
            m.write("data", _host(arrays["data"]))
            res["data"] = tag
        if not self._is_key:
            fk = arrays["feat_key"]
            ref = getattr(fk, "device_ref", None)
            if not (ref and ref[0] is m and ref[1] == "feat"):
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
                out[name] = self._feat_handle(N, H, W)
        return [out]

    def _logits_handle(self, N, H, W, ncls=19):
        m = self._model

        def labels():
            # mx.ndarray.argmax returns float indices; the fused kernel wrote uint8 labels
            return DeviceArray(shape=(N, H, W), fetch=lambda: m.read("labels", (N, H, W), np.uint8).astype(np.float32),
                               device_ref=(m, "labels"))
        return DeviceArray(shape=(N, ncls, H, W), fetch=lambda: m.read("logits", (N, ncls, H, W)),
                           device_ref=(m, "logits"), labels_of=labels)

    def _feat_handle(self, N, H, W):
        m = self._model
        h, w = H // 16, W // 16

        def fetch():
            nhwc = m.read("feat", (N, h, w, 2048))
            return np.ascontiguousarray(nhwc.transpose(0, 3, 1, 2))
        return DeviceArray(shape=(N, 2048, h, w), fetch=fetch, device_ref=(m, "feat"))

    def _upload_feat(self, feat_nchw, N, H, W):
        f = np.asarray(feat_nchw, np.float32)
        if f.shape != (N, 2048, H // 16, W // 16):
            raise ValueError("feat_key shape %s does not match the bound graph" % (f.shape,))
        self._model.write("feat", np.ascontiguousarray(f.transpose(0, 2, 3, 1)))

    def plan_for(self, H, W, N=1):
        plan, lw, _ = self._bind((H, W), N)
        return plan, lw


def _fingerprint(a):
    """(host address, shape, sampled content) of an input array: equal tags <=> the bytes already in HBM are still
    valid.  The address identifies the host buffer (two mx.nd.array() handles of one numpy image share it, as in
    demo.py:176-181); the sample (4096 strided elements) catches in-place edits and recycled addresses."""
    h = a.asnumpy() if hasattr(a, "asnumpy") else np.asarray(a)
    flat = h.reshape(-1)
    step = max(1, flat.size // 4096)
    return (h.ctypes.data, h.shape, str(h.dtype), float(np.asarray(flat[::step], np.float64).sum()), float(flat[-1]))


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
