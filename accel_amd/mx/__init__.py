"""`mx`-shaped facade: the few names of the MXNet Python API that the
reference's dff_deeplab files touch, backed by the accel_amd graph layer and
HIP runtime.  `import accel_amd.mx as mx` lets reference-style symbol code run
unchanged (mx.sym.*, mx.symbol.*, mx.contrib.symbol.DeformableConvolution,
mx.operator.register, mx.io.DataBatch, mx.nd.array, mx.gpu)."""
import types as _types

from . import operator
from . import symbol
from . import symbol as sym
from .ndarray import DataBatch, DeviceArray, array, argmax, zeros

contrib = _types.SimpleNamespace(
    symbol=_types.SimpleNamespace(DeformableConvolution=symbol.DeformableConvolution),
    sym=_types.SimpleNamespace(DeformableConvolution=symbol.DeformableConvolution))
io = _types.SimpleNamespace(DataBatch=DataBatch)
sym.split = symbol.split
nd = _types.SimpleNamespace(array=array, argmax=argmax, zeros=zeros, NDArray=DeviceArray)
ndarray = nd


class Context(object):
    def __init__(self, device_type, device_id=0):
        self.device_type = device_type
        self.device_id = device_id

    def __repr__(self):
        return "%s(%d)" % (self.device_type, self.device_id)


def gpu(i=0):
    return Context("gpu", i)


def cpu(i=0):
    return Context("cpu", i)


def cpu_pinned(i=0):
    """mx.cpu_pinned(): host arrays in page-locked memory (fast, overlappable transfers to the GPU)"""
    return Context("cpu_pinned", i)
