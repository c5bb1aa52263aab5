"""Checkpoint IO: counterparts of lib/utils/load_model.py:4-30,72-93 and
lib/utils/save_model.py:4-18, with a self-contained reader/writer for MXNet's
NDArray-list `.params` format (MXNet itself is not a dependency).

File layout (mx.nd.save of a dict):
    uint64 magic = 0x112, uint64 reserved = 0
    uint64 n_arrays, n x NDArray
    uint64 n_names,  n x (uint64 length, bytes)
NDArray, legacy (MXNet <= 0.11, the reference's commit):
    uint32 ndim, uint32 dims[ndim], int32 dev_type, int32 dev_id, int32 type_flag, raw data
NDArray, V1/V2 (MXNet >= 0.12): uint32 magic 0xF993FAC8 (V1) / 0xF993FAC9 (V2)
    [V2: int32 storage_type], uint32 ndim, int64 dims[ndim], context, type_flag, raw data
NDArray, V3 (MXNet >= 1.6, numpy shape semantics): magic 0xF993FACA, layout of V2 with a signed ndim;
    ndim 0 is a scalar WITH context / type_flag / one element, ndim -1 an unknown shape with nothing after it
    (before V3 ndim 0 means "empty array" and ends the record).
Names carry the `arg:` / `aux:` prefix (save_model.py:15-18).
"""
import struct

import numpy as np

_LIST_MAGIC = 0x112
_V1, _V2, _V3 = 0xF993FAC8, 0xF993FAC9, 0xF993FACA
_DTYPES = {0: np.float32, 1: np.float64, 2: np.float16, 3: np.uint8, 4: np.int32, 5: np.int8, 6: np.int64}
_FLAGS = {np.dtype(v): k for k, v in _DTYPES.items()}


class _Reader(object):
    def __init__(self, buf):
        self.b, self.o = buf, 0

    def take(self, fmt):
        if self.o + struct.calcsize("<" + fmt) > len(self.b):
            raise ValueError("truncated .params file")
        v = struct.unpack_from("<" + fmt, self.b, self.o)
        self.o += struct.calcsize("<" + fmt)
        return v if len(v) > 1 else v[0]

    def raw(self, n):
        v = self.b[self.o:self.o + n]
        if len(v) != n:
            raise ValueError("truncated .params file")
        self.o += n
        return v


def _read_ndarray(r):
    first = r.take("I")
    scalar_ok = False
    if first in (_V1, _V2, _V3):
        if first != _V1:
            stype = r.take("i")
            if stype != 0:
                raise NotImplementedError("sparse NDArray (storage type %d) in checkpoint" % stype)
        ndim = r.take("i") if first == _V3 else r.take("I")
        scalar_ok = first == _V3
        if ndim < 0:
            return np.zeros((0,), np.float32)       # V3 "unknown shape": no payload
        shape = [r.take("q") for _ in range(ndim)]
    else:
        ndim = first
        if ndim > 32:
            raise ValueError("corrupt .params file: array of %d dimensions" % ndim)
        shape = [r.take("I") for _ in range(ndim)]
    if ndim == 0 and not scalar_ok:
        return np.zeros((), np.float32)             # pre-V3: ndim 0 = empty array, record ends here
    r.take("ii")                       # context: dev_type, dev_id (where it was saved from; irrelevant on load)
    flag = r.take("i")
    if flag not in _DTYPES:
        raise ValueError("corrupt or unsupported .params file: type flag %d" % flag)
    dt = np.dtype(_DTYPES[flag])
    n = int(np.prod(shape)) if shape else 1
    return np.frombuffer(r.raw(n * dt.itemsize), dtype=dt).reshape(shape).copy()


def nd_load(path):
    """-> dict name -> numpy array (the `mx.nd.load` of a saved dict)."""
    with open(path, "rb") as f:
        r = _Reader(f.read())
    magic, _ = r.take("QQ")
    if magic != _LIST_MAGIC:
        raise ValueError("%s: not an MXNet NDArray-list file (magic %#x)" % (path, magic))
    n = r.take("Q")
    arrays = [_read_ndarray(r) for _ in range(n)]
    n_names = r.take("Q")
    names = []
    for _ in range(n_names):
        ln = r.take("Q")
        names.append(r.raw(ln).decode())
    if n_names != n:
        raise ValueError("%s: %d arrays but %d names" % (path, n, n_names))
    return dict(zip(names, arrays))


def nd_save(path, data, legacy=True):
    """Writes `data` (dict name -> array) in the NDArray-list format."""
    names = list(data.keys())
    out = [struct.pack("<QQQ", _LIST_MAGIC, 0, len(names))]
    for k in names:
        a = np.ascontiguousarray(data[k])
        if a.dtype not in _FLAGS:
            a = a.astype(np.float32)
        if legacy:
            out.append(struct.pack("<I", a.ndim) + struct.pack("<%dI" % a.ndim, *a.shape))
        else:
            out.append(struct.pack("<IiI", _V2, 0, a.ndim) + struct.pack("<%dq" % a.ndim, *a.shape))
        out.append(struct.pack("<iii", 1, 0, _FLAGS[a.dtype]))
        out.append(a.tobytes())
    out.append(struct.pack("<Q", len(names)))
    for k in names:
        kb = k.encode()
        out.append(struct.pack("<Q", len(kb)) + kb)
    with open(path, "wb") as f:
        f.write(b"".join(out))


def load_checkpoint(prefix, epoch, argprefix=''):
    """load_model.py:4-30: split `arg:` / `aux:`; names lacking `argprefix` get it prepended."""
    return _split(nd_load('%s-%04d.params' % (prefix, epoch)), argprefix)


def _split(save_dict, argprefix=''):
    arg_params, aux_params = {}, {}
    for k, v in save_dict.items():
        tp, name = k.split(':', 1)
        if name[:len(argprefix)] != argprefix:
            name = argprefix + name
        if tp == 'arg':
            arg_params[name] = v
        if tp == 'aux':
            aux_params[name] = v
    return arg_params, aux_params


def load_param(prefix, epoch, convert=False, ctx=None, process=False, argprefix=''):
    """load_model.py:73-93 (`convert`/`ctx` are accepted and ignored: parameters are
    uploaded when a Predictor binds)."""
    arg_params, aux_params = load_checkpoint(prefix, epoch, argprefix)
    if process:
        for test in [k for k in arg_params.keys() if '_test' in k]:
            arg_params[test.replace('_test', '')] = arg_params.pop(test)
    return arg_params, aux_params


def load_param_file(path, process=False, argprefix=''):
    arg_params, aux_params = _split(nd_load(path), argprefix)
    if process:
        for test in [k for k in arg_params.keys() if '_test' in k]:
            arg_params[test.replace('_test', '')] = arg_params.pop(test)
    return arg_params, aux_params


def save_checkpoint(prefix, epoch, arg_params, aux_params):
    """save_model.py:4-18"""
    d = {('arg:%s' % k): v for k, v in arg_params.items()}
    d.update({('aux:%s' % k): v for k, v in aux_params.items()})
    nd_save('%s-%04d.params' % (prefix, epoch), d)
