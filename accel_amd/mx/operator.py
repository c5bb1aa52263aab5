"""The `mx.operator` plugin surface the reference's operator_py modules use
(dff_deeplab/operator_py/tile_as.py:12-50, rpn_inv_normalize.py:12-54):

    @mx.operator.register('<op_type>')
    class XProp(mx.operator.CustomOpProp):
        def __init__(self, **str_kwargs): ...
        def list_arguments(self) / list_outputs(self)
        def infer_shape(self, in_shape) -> (in_shapes, out_shapes[, aux_shapes])
        def create_operator(self, ctx, shapes, dtypes) -> CustomOp
    class X(mx.operator.CustomOp):
        def forward(self, is_train, req, in_data, out_data, aux)
        def backward(...)

and `mx.sym.Custom(<named inputs>, op_type='<op_type>', **params)` at the use
site (accel_18.py:255-256).  Registered operators whose `lowering` attribute
names a fused kernel op are executed by the HIP plan; the others stay
host-side CustomOps, exactly like MXNet runs Python CustomOps on the host.
"""

_REGISTRY = {}


class CustomOp(object):
    def forward(self, is_train, req, in_data, out_data, aux):
        raise NotImplementedError()

    def backward(self, req, out_grad, in_data, out_data, in_grad, aux):
        raise NotImplementedError()

    def assign(self, dst, req, src):
        """MXNet semantics: 'null' -> skip, 'write'/'inplace' -> overwrite, 'add' -> accumulate."""
        if req == "null":
            return
        if req in ("write", "inplace"):
            dst[:] = src
        elif req == "add":
            dst[:] += src
        else:
            raise ValueError("unknown req %r" % (req,))


class CustomOpProp(object):
    def __init__(self, need_top_grad=False):
        self.need_top_grad_ = need_top_grad

    def list_arguments(self):
        return ["data"]

    def list_outputs(self):
        return ["output"]

    def list_auxiliary_states(self):
        return []

    def infer_shape(self, in_shape):
        return in_shape, [in_shape[0]] * len(self.list_outputs()), []

    def infer_type(self, in_type):
        return in_type, [in_type[0]] * len(self.list_outputs()), [in_type[0]] * len(self.list_auxiliary_states())

    def declare_backward_dependency(self, out_grad, in_data, out_data):
        return []

    def create_operator(self, ctx, in_shapes, in_dtypes):
        return CustomOp()


def register(reg_name):
    def do_register(prop_cls):
        if not issubclass(prop_cls, CustomOpProp):
            raise TypeError("register(%r): %r is not a CustomOpProp" % (reg_name, prop_cls))
        _REGISTRY[reg_name] = prop_cls
        return prop_cls
    return do_register


def create_prop(op_type, **params):
    if op_type not in _REGISTRY:
        raise KeyError("Custom operator %r is not registered (registered: %s)"
                       % (op_type, sorted(_REGISTRY)))
    return _REGISTRY[op_type](**params)


def registered():
    return sorted(_REGISTRY)
