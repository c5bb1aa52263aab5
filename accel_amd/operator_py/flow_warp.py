"""`FlowWarp`: the flow-warping operator of the Accel path, registered through
the operator_py plugin surface (same convention as the reference's
dff_deeplab/operator_py/tile_as.py:12-50).

The reference itself has no op of this name: it warps with two stock MXNet ops,
GridGenerator(transform_type='warp') + BilinearSampler (accel_18.py:174-175;
SURVEY.md F1).  `mx.sym.Custom(data=feat, flow=flow, op_type='FlowWarp')` is an
alias for that pair; both spellings lower to the same fused HIP kernel
(accel_amd/csrc/misc.hip flow_warp_kernel).  forward() runs that kernel through
the C ABI (accel_flow_warp) on host arrays -- there is no CPU fallback.
"""
from .. import mx


class FlowWarpOperator(mx.operator.CustomOp):
    def __init__(self, device_id=0):
        super(FlowWarpOperator, self).__init__()
        self._device_id = device_id
        self._ctx = None

    def forward(self, is_train, req, in_data, out_data, aux):
        from .. import runtime
        if self._ctx is None:
            self._ctx = runtime.Context(self._device_id)
        feat, flow = (a.asnumpy() if hasattr(a, "asnumpy") else a for a in in_data[:2])
        self.assign(out_data[0], req[0], self._ctx.flow_warp(feat, flow))

    def backward(self, req, out_grad, in_data, out_data, in_grad, aux):
        raise NotImplementedError("FlowWarp is inference-only on this path")


@mx.operator.register('FlowWarp')
class FlowWarpProp(mx.operator.CustomOpProp):
    lowering = "warp"

    def __init__(self):
        super(FlowWarpProp, self).__init__(need_top_grad=False)

    def list_arguments(self):
        return ['data', 'flow']

    def list_outputs(self):
        return ['output']

    def infer_shape(self, in_shape):
        data_shape, flow_shape = in_shape
        assert flow_shape[1] == 2 and list(flow_shape[2:]) == list(data_shape[2:]), \
            'flow must be (N, 2, H, W) matching data (N, C, H, W)'
        return [data_shape, flow_shape], [data_shape]

    def create_operator(self, ctx, shapes, dtypes):
        return FlowWarpOperator(getattr(ctx, 'device_id', 0))

    def declare_backward_dependency(self, out_grad, in_data, out_data):
        return []
