"""Array handles exchanged between harness code and the Predictor.

A DeviceArray is either host data (numpy) or a reference to a buffer that
lives in HBM inside a plan/model of the HIP runtime; `.asnumpy()` is the only
synchronising call, like `mx.nd.NDArray.asnumpy()` in the reference's demo
loop (dff_deeplab/demo.py:238,245)."""
import numpy as np


class DeviceArray(object):
    def __init__(self, host=None, shape=None, fetch=None, device_ref=None, labels_of=None):
        self._host = None if host is None else np.ascontiguousarray(host)
        self._shape = tuple(shape) if shape is not None else tuple(self._host.shape)
        self._fetch = fetch            # callable -> numpy (blocks)
        self.device_ref = device_ref   # (owner, buffer name) when resident in HBM
        self.labels_of = labels_of     # callable -> DeviceArray of the fused argmax, if any

    @property
    def shape(self):
        return self._shape

    @property
    def on_device(self):
        return self._host is None

    def asnumpy(self):
        if self._host is None:
            self._host = np.ascontiguousarray(self._fetch())
        return self._host

    def __repr__(self):
        return "<DeviceArray %s %s>" % ("x".join(map(str, self._shape)),
                                        "hbm" if self._host is None else "host")


def array(src, ctx=None, dtype=np.float32):
    if isinstance(src, DeviceArray):
        return src
    return DeviceArray(host=np.asarray(src, dtype=dtype))


def zeros(shape, ctx=None, dtype=np.float32):
    return DeviceArray(host=np.zeros(shape, dtype))


def argmax(arr, axis=1):
    """mx.ndarray.argmax(out, axis=1): when `arr` is a logits buffer produced by
    the fused score kernel the label map already exists in HBM (the kernel
    writes logits and first-max labels together); otherwise reduce on the host."""
    if isinstance(arr, DeviceArray) and arr.labels_of is not None and axis == 1:
        return arr.labels_of()
    a = arr.asnumpy() if isinstance(arr, DeviceArray) else np.asarray(arr)
    return DeviceArray(host=np.argmax(a, axis=axis).astype(np.float32))


class DataBatch(object):
    """mx.io.DataBatch as built at demo.py:210-212,229-231."""

    def __init__(self, data, label=None, pad=0, index=None, provide_data=None, provide_label=None):
        self.data = data
        self.label = label
        self.pad = pad
        self.index = index
        self.provide_data = provide_data
        self.provide_label = provide_label
