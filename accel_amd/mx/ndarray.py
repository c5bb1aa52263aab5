"""Array handles exchanged between harness code and the Predictor.

A DeviceArray is either host data (numpy) or a reference to a buffer that
lives in HBM inside a plan/model of the HIP runtime; `.asnumpy()` is the only
synchronising call, like `mx.nd.NDArray.asnumpy()` in the reference's demo
loop (dff_deeplab/demo.py:238,245)."""
import itertools

import numpy as np

_UID = itertools.count(1)


class DeviceArray(object):
    """Immutable once built (like an NDArray the harness never writes into): `uid` identifies the CONTENT, which is what
    lets a Predictor recognise that the bytes of an input are already in HBM.  Host data is a private copy
    (mx.nd.array copies, as MXNet does) and is handed out read-only."""

    def __init__(self, host=None, shape=None, fetch=None, device_ref=None, labels_of=None, pinned=None):
        self.uid = next(_UID)
        self._host = host
        if self._host is not None and self._host.flags.writeable:
            self._host.setflags(write=False)
        self._shape = tuple(shape) if shape is not None else tuple(self._host.shape)
        self._fetch = fetch            # callable -> numpy (blocks)
        self.device_ref = device_ref   # (owner model, buffer name, write generation) when resident in HBM
        self.labels_of = labels_of     # callable -> DeviceArray of the fused argmax, if any
        self.pinned = pinned           # runtime.PinnedBuffer behind `_host` when built with ctx=mx.cpu_pinned()

    @property
    def shape(self):
        return self._shape

    @property
    def on_device(self):
        return self._host is None

    @property
    def has_host_copy(self):
        return self._host is not None

    def asnumpy(self):
        """Blocks until the data is on the host.  The array is read-only: copy it before editing (MXNet returns a
        fresh copy on every call; one shared read-only copy keeps 160 MB logit maps from being duplicated)."""
        if self._host is None:
            h = np.ascontiguousarray(self._fetch())
            h.setflags(write=False)
            self._host = h
        return self._host

    def __repr__(self):
        return "<DeviceArray %s %s>" % ("x".join(map(str, self._shape)),
                                        "hbm" if self._host is None else "host")


def array(src, ctx=None, dtype=np.float32):
    """mx.nd.array: always a COPY of `src` (later edits of `src` do not reach the array).  With
    ctx=mx.cpu_pinned() the copy lives in page-locked memory, the source of overlapped uploads."""
    if isinstance(src, DeviceArray):
        return src
    a = np.asarray(src)
    if ctx is not None and getattr(ctx, "device_type", "") == "cpu_pinned":
        from .. import runtime
        pb = runtime.PinnedBuffer(a.shape, dtype)
        pb.array[...] = a
        return DeviceArray(host=pb.array, pinned=pb)
    return DeviceArray(host=np.array(a, dtype=dtype, order="C", copy=True))


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
