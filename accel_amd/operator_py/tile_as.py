"""`tile_as`: a host-executed plugin operator registered through the operator_py surface, with the interface of the reference's
operator of the same name (dff_deeplab/operator_py/tile_as.py:12-50; used only by the detection batch symbol): arguments
(`data_content`, `data_shape`) in that order, output `data_tiled` = `data_content` tiled along axis 0 by the batch size of
`data_shape` (mx.ndarray.tile(content, reps=(n, 1, 1, 1)): a content batch of b gives n*b rows), no gradient to either input
(`need_top_grad=False`).  Kept as the worked example of an op that stays on the host (plans reject it: `accel_amd.lower` raises
for Custom ops without a device lowering), next to `FlowWarp`, which does lower to a HIP kernel."""
import numpy as np

from .. import mx


def _to_numpy(a):
    return a.asnumpy() if hasattr(a, "asnumpy") else np.asarray(a)


class TileAsOperator(mx.operator.CustomOp):
    def forward(self, is_train, req, in_data, out_data, aux):
        content = _to_numpy(in_data[0])
        reps = (int(in_data[1].shape[0]),) + (1,) * (content.ndim - 1)
        self.assign(out_data[0], req[0], np.tile(content, reps))

    def backward(self, req, out_grad, in_data, out_data, in_grad, aux):
        for k in (0, 1):
            self.assign(in_grad[k], req[k], 0)


@mx.operator.register('tile_as')
class TileAsProp(mx.operator.CustomOpProp):
    def __init__(self):
        mx.operator.CustomOpProp.__init__(self, need_top_grad=False)

    def list_arguments(self):
        return ['data_content', 'data_shape']

    def list_outputs(self):
        return ['data_tiled']

    def infer_shape(self, in_shape):
        content, shape_like = in_shape
        return [content, shape_like], [[shape_like[0] * content[0]] + list(content[1:])]

    def create_operator(self, ctx, shapes, dtypes):
        return TileAsOperator()
