"""`tile_as`: a host-executed plugin operator registered through the
operator_py surface.  Behaviour of the reference's operator of the same name
(dff_deeplab/operator_py/tile_as.py:12-50, used only by the detection batch
symbol): output = `data_content` repeated along axis 0 up to the batch size of
`data_shape`; the gradient of the content is the sum over that axis, the shape
input gets no gradient.  Kept as the worked example of an op that stays on the
host (plans reject it: `accel_amd.lower` raises for Custom ops without a device
lowering), next to `FlowWarp`, which does lower to a HIP kernel."""
import numpy as np

from .. import mx


def _to_numpy(a):
    return a.asnumpy() if hasattr(a, "asnumpy") else np.asarray(a)


class TileAsOperator(mx.operator.CustomOp):
    def forward(self, is_train, req, in_data, out_data, aux):
        batch = in_data[0].shape[0]
        content = _to_numpy(in_data[1])
        self.assign(out_data[0], req[0], np.repeat(content, batch, axis=0) if content.shape[0] == 1
                    else np.tile(content, (batch,) + (1,) * (content.ndim - 1)))

    def backward(self, req, out_grad, in_data, out_data, in_grad, aux):
        self.assign(in_grad[0], req[0], 0)
        self.assign(in_grad[1], req[1], _to_numpy(out_grad[0]).sum(axis=0, keepdims=True))


@mx.operator.register('tile_as')
class TileAsProp(mx.operator.CustomOpProp):
    def __init__(self):
        mx.operator.CustomOpProp.__init__(self, need_top_grad=True)

    def list_arguments(self):
        return ['data_shape', 'data_content']

    def list_outputs(self):
        return ['output']

    def infer_shape(self, in_shape):
        shape_like, content = in_shape
        return [shape_like, content], [[shape_like[0]] + list(content[1:])]

    def create_operator(self, ctx, shapes, dtypes):
        return TileAsOperator()
