"""`tile_as`: host-side CustomOp, behaviourally the same as the reference's
dff_deeplab/operator_py/tile_as.py:12-50 (tile `data_content` along axis 0 to the
batch size of `data_shape`).  Only the detection batch symbol uses it; it is kept
because the reference's symbol modules register it at import (accel_18.py:11-15)
and as the worked example of a host-executed plugin op."""
import numpy as np

from .. import mx


class TileAsOperator(mx.operator.CustomOp):
    def forward(self, is_train, req, in_data, out_data, aux):
        content = in_data[0].asnumpy() if hasattr(in_data[0], "asnumpy") else np.asarray(in_data[0])
        n = in_data[1].shape[0]
        reps = (n,) + (1,) * (content.ndim - 1)
        self.assign(out_data[0], req[0], np.tile(content, reps))

    def backward(self, req, out_grad, in_data, out_data, in_grad, aux):
        g = out_grad[0].asnumpy() if hasattr(out_grad[0], "asnumpy") else np.asarray(out_grad[0])
        self.assign(in_grad[0], req[0], g.sum(axis=0, keepdims=True))
        self.assign(in_grad[1], req[1], 0)


@mx.operator.register('tile_as')
class TileAsProp(mx.operator.CustomOpProp):
    def __init__(self):
        super(TileAsProp, self).__init__(need_top_grad=True)

    def list_arguments(self):
        return ['data_shape', 'data_content']

    def list_outputs(self):
        return ['output']

    def infer_shape(self, in_shape):
        data_shape, data_content = in_shape
        out = [data_shape[0]] + list(data_content[1:])
        return [data_shape, data_content], [out]

    def create_operator(self, ctx, shapes, dtypes):
        return TileAsOperator()
