"""Graph-description layer with the slice of the `mx.symbol` surface that the
reference's dff_deeplab symbol files use (dff_deeplab/symbols/*.py).

It only DESCRIBES a network: nodes, attributes, parameter names and shapes.
Nothing here computes; `accel_amd.lower` turns a described graph into a fused
kernel plan that the HIP executor (accel_amd/csrc) runs.

Naming rules follow MXNet so that reference checkpoints line up
(SURVEY.md appendix A): an op created with name=X owns `X_weight`, `X_bias`,
`X_gamma`, `X_beta` (args) and `X_moving_mean`, `X_moving_var` (aux states);
its output is called `X_output`; a Variable's output is its own name.
"""
import itertools
import math

_COUNTERS = {}


def _auto_name(hint):
    i = _COUNTERS.get(hint, 0)
    _COUNTERS[hint] = i + 1
    return "%s%d" % (hint, i)


def _pair(v, default=None):
    if v is None:
        return default
    if isinstance(v, (int, float)):
        return (int(v), int(v))
    return (int(v[0]), int(v[1]))


class Symbol(object):
    """One node output (or a group of them)."""

    def __init__(self, op, name, inputs=(), attrs=None, aux=(), group=None):
        self.op = op                  # 'null' for variables, '_group' for Group
        self.name = name
        self.inputs = list(inputs)    # Symbols
        self.attrs = dict(attrs or {})
        self.aux = list(aux)          # aux-state Variables (BatchNorm only)
        self.group = group            # list of Symbols for '_group'
        self.is_aux = False

    # -- arithmetic sugar used by get_flownet (`img / 255.0`, `conv * 2.5`) and
    #    residual_unit (`conv3 + shortcut`)
    def __truediv__(self, s):
        return Symbol("_div_scalar", _auto_name("_divscalar"), [self], {"scalar": float(s)})

    __div__ = __truediv__

    def __mul__(self, s):
        if isinstance(s, Symbol):
            raise NotImplementedError("symbol * symbol is not on the Accel path")
        return Symbol("_mul_scalar", _auto_name("_mulscalar"), [self], {"scalar": float(s)})

    def __add__(self, other):
        if not isinstance(other, Symbol):
            raise NotImplementedError("symbol + scalar is not on the Accel path")
        return Symbol("elemwise_add", _auto_name("_plus"), [self, other])

    # -- graph queries -------------------------------------------------------
    def _heads(self):
        return self.group if self.op == "_group" else [self]

    def topo(self):
        """Nodes in MXNet's DFS post-order (inputs before aux, as nnvm does)."""
        seen, order = set(), []

        def visit(s):
            if id(s) in seen:
                return
            seen.add(id(s))
            for i in s.inputs:
                visit(i)
            for a in s.aux:
                visit(a)
            order.append(s)

        for h in self._heads():
            visit(h)
        return order

    def list_arguments(self):
        return [s.name for s in self.topo() if s.op == "null" and not s.is_aux]

    def list_auxiliary_states(self):
        return [s.name for s in self.topo() if s.op == "null" and s.is_aux]

    def list_outputs(self):
        return [h.name if h.op == "null" else h.name + "_output" for h in self._heads()]

    def get_internals(self):
        return Group([s for s in self.topo()])

    def infer_shape(self, **known):
        """-> (arg_shapes, out_shapes, aux_shapes) ordered like list_*()."""
        shapes = infer_shapes(self, known)
        args = [shapes[id(s)] for s in self.topo() if s.op == "null" and not s.is_aux]
        auxs = [shapes[id(s)] for s in self.topo() if s.op == "null" and s.is_aux]
        outs = [shapes[id(h)] for h in self._heads()]
        return args, outs, auxs

    def __repr__(self):
        return "<Symbol %s %s>" % (self.op, self.name)


def Variable(name, **attrs):
    return Symbol("null", name, attrs=attrs)


def Group(symbols):
    return Symbol("_group", "group", group=list(symbols))


def _var_or(given, name, **attrs):
    return given if given is not None else Variable(name, **attrs)


def Convolution(data=None, weight=None, bias=None, kernel=None, stride=None, dilate=None,
                pad=None, num_filter=None, num_group=1, no_bias=False, name=None,
                workspace=None, cudnn_off=None, attr=None, **_):
    name = name or _auto_name("convolution")
    ins = [data, _var_or(weight, name + "_weight")]
    if not no_bias:
        ins.append(_var_or(bias, name + "_bias"))
    return Symbol("Convolution", name, ins, {
        "kernel": _pair(kernel), "stride": _pair(stride, (1, 1)), "dilate": _pair(dilate, (1, 1)),
        "pad": _pair(pad, (0, 0)), "num_filter": int(num_filter), "num_group": int(num_group),
        "no_bias": bool(no_bias)})


def Deconvolution(data=None, weight=None, bias=None, kernel=None, stride=None, pad=None,
                  num_filter=None, num_group=1, no_bias=True, name=None, workspace=None,
                  attr=None, **_):
    name = name or _auto_name("deconvolution")
    ins = [data, _var_or(weight, name + "_weight")]
    if not no_bias:
        ins.append(_var_or(bias, name + "_bias"))
    return Symbol("Deconvolution", name, ins, {
        "kernel": _pair(kernel), "stride": _pair(stride, (1, 1)), "pad": _pair(pad, (0, 0)),
        "num_filter": int(num_filter), "num_group": int(num_group), "no_bias": bool(no_bias)})


def DeformableConvolution(data=None, offset=None, weight=None, bias=None, kernel=None,
                          stride=None, dilate=None, pad=None, num_filter=None, num_group=1,
                          num_deformable_group=1, no_bias=False, name=None, **_):
    name = name or _auto_name("deformableconvolution")
    ins = [data, offset, _var_or(weight, name + "_weight")]
    if not no_bias:
        ins.append(_var_or(bias, name + "_bias"))
    return Symbol("DeformableConvolution", name, ins, {
        "kernel": _pair(kernel), "stride": _pair(stride, (1, 1)), "dilate": _pair(dilate, (1, 1)),
        "pad": _pair(pad, (0, 0)), "num_filter": int(num_filter), "num_group": int(num_group),
        "num_deformable_group": int(num_deformable_group), "no_bias": bool(no_bias)})


def BatchNorm(data=None, gamma=None, beta=None, eps=1e-3, momentum=0.9, fix_gamma=True,
              use_global_stats=False, name=None, **_):
    name = name or _auto_name("batchnorm")
    mean = Variable(name + "_moving_mean")
    var = Variable(name + "_moving_var")
    mean.is_aux = var.is_aux = True
    return Symbol("BatchNorm", name,
                  [data, _var_or(gamma, name + "_gamma"), _var_or(beta, name + "_beta")],
                  {"eps": float(eps), "fix_gamma": bool(fix_gamma),
                   "use_global_stats": bool(use_global_stats)}, aux=[mean, var])


def Activation(data=None, act_type="relu", name=None, **_):
    return Symbol("Activation", name or _auto_name("activation"), [data], {"act_type": act_type})


def LeakyReLU(data=None, act_type="leaky", slope=0.25, name=None, **_):
    return Symbol("LeakyReLU", name or _auto_name("leakyrelu"), [data],
                  {"act_type": act_type, "slope": float(slope)})


def Pooling(data=None, kernel=None, stride=None, pad=None, pool_type="max",
            pooling_convention="valid", global_pool=False, name=None, **_):
    return Symbol("Pooling", name or _auto_name("pooling"), [data], {
        "kernel": _pair(kernel), "stride": _pair(stride, (1, 1)), "pad": _pair(pad, (0, 0)),
        "pool_type": pool_type, "pooling_convention": pooling_convention,
        "global_pool": bool(global_pool)})


def Concat(*data, **kw):
    """dim=1: channel concatenation; dim=0: batch concatenation (the training graphs stack frames: accel_18.py:47-48)"""
    name = kw.get("name") or _auto_name("concat")
    dim = int(kw.get("dim", 1))
    if dim not in (0, 1):
        raise NotImplementedError("Concat(dim=%r)" % (dim,))
    return Symbol("Concat", name, list(data), {"dim": dim})


class _SplitOutputs(list):
    """mx.sym.split(...) result: indexable, one Symbol per slice (accel_18.py:43,54)."""


def split(data=None, num_outputs=None, axis=1, squeeze_axis=False, name=None, **_):
    """SliceChannel: `num_outputs` equal slices along `axis` (only the batch axis occurs on the Accel path)."""
    if axis != 0 or squeeze_axis:
        raise NotImplementedError("split(axis=%r, squeeze_axis=%r)" % (axis, squeeze_axis))
    name = name or _auto_name("split")
    n = int(num_outputs)
    return _SplitOutputs(Symbol("_split_out", "%s_output%d" % (name, i) if n > 1 else name, [data],
                                {"index": i, "num_outputs": n, "axis": 0}) for i in range(n))


SliceChannel = split


def Crop(*data, **kw):
    name = kw.get("name") or _auto_name("crop")
    return Symbol("Crop", name, list(data), {
        "offset": _pair(kw.get("offset"), (0, 0)), "h_w": _pair(kw.get("h_w"), (0, 0)),
        "center_crop": bool(kw.get("center_crop", False)), "num_args": len(data)})


def broadcast_add(*data, **kw):
    lhs = kw.get("lhs", data[0] if data else None)
    rhs = kw.get("rhs", data[1] if len(data) > 1 else None)
    return Symbol("broadcast_add", kw.get("name") or _auto_name("broadcast_add"), [lhs, rhs])


def GridGenerator(data=None, transform_type="affine", target_shape=None, name=None, **_):
    if transform_type != "warp":
        raise NotImplementedError("GridGenerator(transform_type=%r)" % transform_type)
    return Symbol("GridGenerator", name or _auto_name("gridgenerator"), [data],
                  {"transform_type": "warp"})


def BilinearSampler(data=None, grid=None, name=None, **_):
    return Symbol("BilinearSampler", name or _auto_name("bilinearsampler"), [data, grid])


def SoftmaxOutput(data=None, label=None, name=None, **attrs):
    name = name or _auto_name("softmaxoutput")
    return Symbol("SoftmaxOutput", name, [data, _var_or(label, name + "_label")], attrs)


def Custom(*data, **kw):
    """mx.sym.Custom(<inputs>, op_type=..., **params): dispatches to an
    operator registered through accel_amd.mx.operator.register (the reference's
    operator_py convention, dff_deeplab/operator_py/tile_as.py:12-50)."""
    from . import operator as _op
    op_type = kw.pop("op_type")
    name = kw.pop("name", None) or _auto_name(op_type.lower())
    named = {k: v for k, v in kw.items() if isinstance(v, Symbol)}
    params = {k: str(v) for k, v in kw.items() if not isinstance(v, Symbol)}
    prop = _op.create_prop(op_type, **params)
    ins = list(data) + [named[a] for a in prop.list_arguments() if a in named]
    if len(ins) != len(prop.list_arguments()):
        raise ValueError("Custom(%s): expected inputs %s" % (op_type, prop.list_arguments()))
    return Symbol("Custom", name, ins, {"op_type": op_type, "params": params, "prop": prop})


# ---------------------------------------------------------------------------
# shape inference
# ---------------------------------------------------------------------------
def _conv_out(n, k, s, p, d):
    return (n + 2 * p - d * (k - 1) - 1) // s + 1


def _pool_out(n, k, s, p, full):
    if full:
        return 1 + int(math.ceil(float(n + 2 * p - k) / s))
    return 1 + (n + 2 * p - k) // s


def infer_shapes(sym, known):
    """Forward shape propagation; parameter shapes are deduced from their
    consumer like MXNet's bidirectional inference does for these ops."""
    sh = {}
    nodes = sym.topo()
    for s in nodes:
        if s.op == "null" and s.name in known:
            sh[id(s)] = tuple(int(v) for v in known[s.name])

    def need(x):
        if id(x) not in sh:
            raise ValueError("cannot infer shape of %r (missing input shape?)" % (x,))
        return sh[id(x)]

    def setp(x, shape):
        if x.op == "null":
            if id(x) in sh and sh[id(x)] != tuple(shape):
                raise ValueError("shape mismatch for %s: %s vs %s" % (x.name, sh[id(x)], shape))
            sh[id(x)] = tuple(shape)

    for s in nodes:
        op, a = s.op, s.attrs
        if op == "null":
            continue
        if op in ("Convolution", "DeformableConvolution"):
            n, c, h, w = need(s.inputs[0])
            kh, kw = a["kernel"]
            g = a["num_group"]
            widx = 1 if op == "Convolution" else 2
            setp(s.inputs[widx], (a["num_filter"], c // g, kh, kw))
            if not a["no_bias"]:
                setp(s.inputs[widx + 1], (a["num_filter"],))
            ho = _conv_out(h, kh, a["stride"][0], a["pad"][0], a["dilate"][0])
            wo = _conv_out(w, kw, a["stride"][1], a["pad"][1], a["dilate"][1])
            if op == "DeformableConvolution":
                off = need(s.inputs[1])
                exp = (n, 2 * kh * kw * a["num_deformable_group"], ho, wo)
                if off != exp:
                    raise ValueError("%s: offset shape %s, expected %s" % (s.name, off, exp))
            sh[id(s)] = (n, a["num_filter"], ho, wo)
        elif op == "Deconvolution":
            n, c, h, w = need(s.inputs[0])
            kh, kw = a["kernel"]
            g = a["num_group"]
            setp(s.inputs[1], (c, a["num_filter"] // g, kh, kw))
            if not a["no_bias"]:
                setp(s.inputs[2], (a["num_filter"],))
            sh[id(s)] = (n, a["num_filter"], a["stride"][0] * (h - 1) + kh - 2 * a["pad"][0],
                         a["stride"][1] * (w - 1) + kw - 2 * a["pad"][1])
        elif op == "BatchNorm":
            shp = need(s.inputs[0])
            for p in s.inputs[1:] + s.aux:
                setp(p, (shp[1],))
            sh[id(s)] = shp
        elif op in ("Activation", "LeakyReLU", "_div_scalar", "_mul_scalar", "GridGenerator"):
            sh[id(s)] = need(s.inputs[0])
        elif op == "Pooling":
            n, c, h, w = need(s.inputs[0])
            full = a["pooling_convention"] == "full"
            sh[id(s)] = (n, c, _pool_out(h, a["kernel"][0], a["stride"][0], a["pad"][0], full),
                         _pool_out(w, a["kernel"][1], a["stride"][1], a["pad"][1], full))
        elif op == "Concat" and a["dim"] == 0:
            shps = [need(i) for i in s.inputs]
            for t in shps[1:]:
                if t[1:] != shps[0][1:]:
                    raise ValueError("Concat %s (dim 0): incompatible shapes %s" % (s.name, shps))
            sh[id(s)] = (sum(t[0] for t in shps),) + shps[0][1:]
        elif op == "Concat":
            shps = [need(i) for i in s.inputs]
            for t in shps[1:]:
                if t[0] != shps[0][0] or t[2:] != shps[0][2:]:
                    raise ValueError("Concat %s: incompatible shapes %s" % (s.name, shps))
            sh[id(s)] = (shps[0][0], sum(t[1] for t in shps)) + shps[0][2:]
        elif op == "_split_out":
            src = need(s.inputs[0])
            if src[0] % a["num_outputs"]:
                raise ValueError("split %s: batch %d is not divisible by %d" % (s.name, src[0], a["num_outputs"]))
            sh[id(s)] = (src[0] // a["num_outputs"],) + src[1:]
        elif op == "Crop":
            src = need(s.inputs[0])
            if a["num_args"] == 2:
                ref = need(s.inputs[1])
                hw = ref[2:]
            else:
                hw = a["h_w"]
            oy, ox = a["offset"]
            if oy + hw[0] > src[2] or ox + hw[1] > src[3]:
                raise ValueError("Crop %s: %s does not fit in %s at offset %s" % (s.name, hw, src, a["offset"]))
            sh[id(s)] = src[:2] + tuple(hw)
        elif op in ("broadcast_add", "elemwise_add"):
            l, r = need(s.inputs[0]), need(s.inputs[1])
            if l != r:
                raise ValueError("%s %s: %s vs %s" % (op, s.name, l, r))
            sh[id(s)] = l
        elif op == "BilinearSampler":
            d, g = need(s.inputs[0]), need(s.inputs[1])
            sh[id(s)] = d[:2] + g[2:]
        elif op == "SoftmaxOutput":
            d = need(s.inputs[0])
            if s.attrs.get("multi_output"):
                setp(s.inputs[1], (d[0],) + tuple(d[2:]))
            else:
                setp(s.inputs[1], (d[0],))
            sh[id(s)] = d
        elif op == "Custom":
            ins = [list(need(i)) for i in s.inputs]
            res = a["prop"].infer_shape(ins)
            sh[id(s)] = tuple(res[1][0])
        elif op == "_group":
            continue
        else:
            raise NotImplementedError("infer_shape for op %s" % op)
    return sh
