"""ctypes binding of libaccel_hip.so (include/accel_hip.h).

The library is the only compute path of accel_amd: there is no CPU fallback.
If it is missing or no MI355X is visible, every call raises AccelError.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("ACCEL_LIB_PATH") or os.path.join(_HERE, "libaccel_hip.so")      # (override: A/B of two builds)
_lib = None


class AccelError(RuntimeError):
    pass


def build(force=False):
    """Compile the HIP extension in-tree for gfx950 (hipcc cross-compiles without a GPU)."""
    src = os.path.join(_HERE, "csrc")
    stale = force or not os.path.exists(LIB_PATH)
    if not stale:
        t = os.path.getmtime(LIB_PATH)
        for f in os.listdir(src):
            if f.endswith((".hip", ".cpp", ".h")) and os.path.getmtime(os.path.join(src, f)) > t:
                stale = True
        hdr = os.path.join(_HERE, "..", "include", "accel_hip.h")
        if os.path.exists(hdr) and os.path.getmtime(hdr) > t:
            stale = True
    if stale:
        subprocess.check_call(["make", "-C", src, "-j4"], stdout=subprocess.DEVNULL)
    return LIB_PATH


def _declare(lib):
    c = ctypes
    vp, i, f, sz = c.c_void_p, c.c_int, c.c_float, c.c_size_t
    lib.accel_last_error.restype = c.c_char_p
    lib.accel_version.restype = c.c_char_p
    lib.accel_ctx_stream.restype = vp
    lib.accel_ctx_stream.argtypes = [vp]
    sigs = {
        "accel_ctx_create": [i, c.POINTER(vp)],
        "accel_ctx_destroy": [vp],
        "accel_sync": [vp],
        "accel_model_create": [vp, c.POINTER(vp)],
        "accel_model_destroy": [vp],
        "accel_model_set_param": [vp, c.c_char_p, vp, i, c.POINTER(c.c_int64)],
        "accel_model_has_param": [vp, c.c_char_p],
        "accel_model_add_plan": [vp, c.c_char_p, c.c_char_p, c.POINTER(vp)],
        "accel_plan_op_launch": [vp, c.c_int, c.POINTER(c.c_int), c.POINTER(c.c_int), c.POINTER(c.c_int)],
        "accel_plan_op_mode": [vp, c.c_int, c.POINTER(c.c_int)],
        "accel_plan_op_range": [vp, c.c_int, c.POINTER(c.c_float), c.POINTER(c.c_int)],
        "accel_plan_op_range_words": [vp, c.c_int, vp, c.c_int],
        "accel_tune_stats": [c.POINTER(c.c_int), c.POINTER(c.c_int), c.POINTER(c.c_int)],
        "accel_plan_finalize": [vp],
        "accel_plan_run": [vp],
        "accel_plan_num_ops": [vp],
        "accel_plan_op_info": [vp, i, c.c_char_p, c.c_char_p, c.POINTER(c.c_double), c.POINTER(c.c_double)],
        "accel_plan_profile": [vp, i, vp, i],
        "accel_plan_run_serial": [vp],
        "accel_plan_arena_read": [vp, sz, vp, sz, c.POINTER(sz)],
        "accel_model_write": [vp, c.c_char_p, vp, sz, i],
        "accel_model_bind_device": [vp, c.c_char_p, vp, sz],
        "accel_model_read": [vp, c.c_char_p, vp, sz, i],
        "accel_model_buffer": [vp, c.c_char_p, c.POINTER(vp), c.POINTER(sz)],
        "accel_model_buffer_generation": [vp, c.c_char_p, c.POINTER(c.c_uint64)],
        "accel_host_alloc": [sz, c.POINTER(vp)],
        "accel_host_free": [vp],
        "accel_model_prefetch": [vp, c.c_char_p, vp, sz],
        "accel_model_commit": [vp, c.c_char_p],
        "accel_model_read_async": [vp, c.c_char_p, vp, sz],
        "accel_comm_available": [],
        "accel_comm_unique_id": [vp],
        "accel_comm_create": [vp, i, i, vp, c.POINTER(vp)],
        "accel_comm_destroy": [vp],
        "accel_gather_logits": [vp, vp, vp, sz, i],
        "accel_gather_frames": [vp, vp, sz, vp, sz, i],
        "accel_expand_scores": [vp, vp, i, vp, vp, vp],
        "accel_gather_scores": [vp, vp, i, i, vp, vp, vp, i],
        "accel_comm_sync": [vp],
        "accel_key_forward": [vp, vp, i, vp, vp, vp, i],
        "accel_cur_forward": [vp, vp, vp, i, vp, vp, vp, i],
        "accel_conv2d": [vp, vp, i, i, i, i, vp, vp, i, i, i, i, i, i, i, i, i, vp, vp, vp, i, f, i, vp],
        "accel_deconv2d_4x4s2": [vp, vp, i, i, i, i, vp, vp, i, i, f, vp],
        "accel_deform_conv2d": [vp, vp, i, i, i, i, vp, vp, i, i, i, i, i, i, i, i, i, i, vp],
        "accel_pool2d": [vp, vp, i, i, i, i, i, i, i, i, i, i, i, i, vp, vp, i, vp],
        "accel_flow_warp": [vp, vp, i, i, i, vp, vp],
        "accel_score_fuse": [vp, vp, vp, i, i, i, vp, vp, vp, vp, vp, vp],
        "accel_argmax_c": [vp, vp, i, i, i, vp],
        "accel_flow_input": [vp, vp, vp, i, i, vp],
    }
    for name, args in sigs.items():
        fn = getattr(lib, name)
        fn.argtypes = args
        fn.restype = i
    return sigs


EXPORTS = None


def _share_torch_hip_runtime():
    """PyTorch-ROCm wheels bundle their own libamdhip64 (same soname as the system's).  Whichever copy a process loads
    first serves both users; when the SYSTEM copy came first, torch's later CUDA initialisation failed on this image
    ("No HIP GPUs are available": torch 2.10+rocm7.0 against the ROCm 7.2 runtime), while libaccel_hip runs fine on
    torch's copy.  So if torch is installed but not imported yet, its runtime is loaded before libaccel_hip.so --
    located through the import machinery, torch itself is NOT imported -- and any order of use works
    (tests/test_configs_gpu.py::test_torch_initialises_after_the_library).  ACCEL_SYSTEM_HIP=1 skips this."""
    import sys
    if "torch" in sys.modules or os.environ.get("ACCEL_SYSTEM_HIP") == "1":
        return
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
        if spec is None or not spec.origin:
            return
        cand = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
        if os.path.exists(cand):
            ctypes.CDLL(cand, mode=ctypes.RTLD_GLOBAL)
    except Exception:
        pass


def lib():
    global _lib, EXPORTS
    if _lib is None:
        _share_torch_hip_runtime()
        if not os.path.exists(LIB_PATH):
            raise AccelError("libaccel_hip.so is not built (run `python -c 'import __graft_entry__ as g; g.build()'`); "
                             "accel_amd has no CPU fallback")
        _lib = ctypes.CDLL(LIB_PATH)
        EXPORTS = _declare(_lib)
    return _lib


def tune_stats():
    """(decisions replayed from a table, decisions taken by timing, entries of the shipped table) of this process"""
    a, b, c_ = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
    lib().accel_tune_stats(ctypes.byref(a), ctypes.byref(b), ctypes.byref(c_))
    return a.value, b.value, c_.value


def check(rc):
    if rc != 0:
        raise AccelError("libaccel_hip: %s (code %d)" % (lib().accel_last_error().decode(), rc))


def _fp(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


class Context(object):
    def __init__(self, device_id=0):
        self.handle = ctypes.c_void_p()
        check(lib().accel_ctx_create(int(device_id), ctypes.byref(self.handle)))
        self.device_id = int(device_id)

    def sync(self):
        check(lib().accel_sync(self.handle))

    @property
    def stream(self):
        return lib().accel_ctx_stream(self.handle)

    def close(self):
        if self.handle:
            lib().accel_ctx_destroy(self.handle)
            self.handle = ctypes.c_void_p()

    # ---- operator level (host NCHW fp32 in/out) --------------------------------
    def conv2d(self, x, w, bias=None, stride=1, pad=0, dilate=1, scale=None, shift=None,
               residual=None, act=0, slope=0.1, tile=-1):
        x, w = _f32(x), _f32(w)
        N, C, H, W = x.shape
        K, _, kh, kw = w.shape
        p2 = lambda v: (v, v) if np.isscalar(v) else tuple(v)
        (sh, sw), (ph, pw), (dh, dw) = p2(stride), p2(pad), p2(dilate)
        Ho = (H + 2 * ph - dh * (kh - 1) - 1) // sh + 1
        Wo = (W + 2 * pw - dw * (kw - 1) - 1) // sw + 1
        y = np.empty((N, K, Ho, Wo), np.float32)
        keep = [None if a is None else _f32(a) for a in (bias, scale, shift, residual)]
        ptr = [None if a is None else _fp(a) for a in keep]
        check(lib().accel_conv2d(self.handle, _fp(x), N, C, H, W, _fp(w), ptr[0], K, kh, kw, sh, sw, ph, pw,
                                 dh, dw, ptr[1], ptr[2], ptr[3], int(act), float(slope), int(tile), _fp(y)))
        return y

    def deconv2d_4x4s2(self, x, w, bias=None, act=0, slope=0.1):
        x, w = _f32(x), _f32(w)
        N, C, H, W = x.shape
        K = w.shape[1]
        y = np.empty((N, K, 2 * H, 2 * W), np.float32)
        b = None if bias is None else _f32(bias)
        check(lib().accel_deconv2d_4x4s2(self.handle, _fp(x), N, C, H, W, _fp(w), None if b is None else _fp(b),
                                         K, int(act), float(slope), _fp(y)))
        return y

    def deform_conv2d(self, x, offset, w, stride=1, pad=0, dilate=1, dg=1):
        x, offset, w = _f32(x), _f32(offset), _f32(w)
        N, C, H, W = x.shape
        K, _, kh, kw = w.shape
        Ho = (H + 2 * pad - dilate * (kh - 1) - 1) // stride + 1
        Wo = (W + 2 * pad - dilate * (kw - 1) - 1) // stride + 1
        y = np.empty((N, K, Ho, Wo), np.float32)
        check(lib().accel_deform_conv2d(self.handle, _fp(x), N, C, H, W, _fp(offset), _fp(w), K, kh, kw,
                                        stride, stride, pad, pad, dilate, dilate, dg, _fp(y)))
        return y

    def pool2d(self, x, kind, kernel, stride, pad=0, convention="valid", scale=None, shift=None, relu=False):
        x = _f32(x)
        N, C, H, W = x.shape
        full = convention == "full"
        po = lambda n: (1 + -(-(n + 2 * pad - kernel) // stride)) if full else (1 + (n + 2 * pad - kernel) // stride)
        y = np.empty((N, C, po(H), po(W)), np.float32)
        s = None if scale is None else _f32(scale)
        b = None if shift is None else _f32(shift)
        check(lib().accel_pool2d(self.handle, _fp(x), N, C, H, W, int(kind == "max"), int(full), kernel, kernel,
                                 stride, stride, pad, pad, None if s is None else _fp(s),
                                 None if b is None else _fp(b), int(relu), _fp(y)))
        return y

    def flow_warp(self, feat, flow):
        feat, flow = _f32(feat), _f32(flow)
        _, C, H, W = feat.shape
        out = np.empty_like(feat)
        check(lib().accel_flow_warp(self.handle, _fp(feat), C, H, W, _fp(flow), _fp(out)))
        return out

    def score_fuse(self, left, wl, right=None, wr=None, cw=None, cb=None):
        left, wl = _f32(left), _f32(wl)
        _, ncls, Hs, Ws = left.shape
        logits = np.empty((1, ncls, 16 * Hs, 16 * Ws), np.float32)
        labels = np.empty((1, 16 * Hs, 16 * Ws), np.uint8)
        opt = [None if a is None else _f32(a) for a in (right, wr, cw, cb)]
        ptr = [None if a is None else _fp(a) for a in opt]
        check(lib().accel_score_fuse(self.handle, _fp(left), ptr[0], ncls, Hs, Ws, _fp(wl), ptr[1], ptr[2], ptr[3],
                                     _fp(logits), _fp(labels)))
        return logits, labels

    def argmax_c(self, logits):
        logits = _f32(logits)
        _, C, H, W = logits.shape
        labels = np.empty((1, H, W), np.uint8)
        check(lib().accel_argmax_c(self.handle, _fp(logits), C, H, W, _fp(labels)))
        return labels

    def flow_input(self, cur, prev):
        cur, prev = _f32(cur), _f32(prev)
        _, _, H, W = cur.shape
        out = np.empty((1, 6, H // 2, W // 2), np.float32)
        check(lib().accel_flow_input(self.handle, _fp(cur), _fp(prev), H, W, _fp(out)))
        return out


class PinnedBuffer(object):
    """Page-locked host memory (accel_host_alloc) seen as a numpy array: the source of accel_model_prefetch and the
    destination of accel_model_read_async.  Freed with the object."""

    def __init__(self, shape, dtype=np.float32):
        self.shape, self.dtype = tuple(int(v) for v in shape), np.dtype(dtype)
        self.nbytes = int(np.prod(self.shape)) * self.dtype.itemsize
        self._ptr = ctypes.c_void_p()
        check(lib().accel_host_alloc(max(self.nbytes, 1), ctypes.byref(self._ptr)))
        raw = (ctypes.c_char * max(self.nbytes, 1)).from_address(self._ptr.value)
        self.array = np.frombuffer(raw, dtype=self.dtype, count=int(np.prod(self.shape))).reshape(self.shape)

    @property
    def ptr(self):
        return self._ptr

    def close(self):
        if self._ptr:
            self.array = None
            lib().accel_host_free(self._ptr)
            self._ptr = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Comm(object):
    """RCCL communicator of the C ABI (accel_comm_*): one per process/GPU; `gather` is accel_gather_logits."""

    @staticmethod
    def available():
        """None if this process can resolve librccl and the entry points of the gather, else the reason (no communicator is made)"""
        rc = lib().accel_comm_available()
        return None if rc == 0 else lib().accel_last_error().decode()

    @staticmethod
    def unique_id():
        buf = ctypes.create_string_buffer(128)
        check(lib().accel_comm_unique_id(buf))
        return buf.raw

    def __init__(self, ctx, rank, nranks, unique_id):
        self.ctx, self.rank, self.nranks = ctx, int(rank), int(nranks)
        self.handle = ctypes.c_void_p()
        idb = ctypes.create_string_buffer(bytes(unique_id), 128)
        check(lib().accel_comm_create(ctx.handle, self.rank, self.nranks, idb, ctypes.byref(self.handle)))

    def gather(self, send_ptr, recv_ptr, nbytes, root=0, send_bytes=None):
        """send_bytes < nbytes: a root that contributes fewer clips than the slot size (accel_gather_frames)"""
        if send_bytes is None or int(send_bytes) == int(nbytes):
            check(lib().accel_gather_logits(self.handle, ctypes.c_void_p(send_ptr), ctypes.c_void_p(recv_ptr) if recv_ptr else None,
                                            int(nbytes), int(root)))
        else:
            check(lib().accel_gather_frames(self.handle, ctypes.c_void_p(send_ptr), int(send_bytes),
                                            ctypes.c_void_p(recv_ptr) if recv_ptr else None, int(nbytes), int(root)))

    def gather_scores(self, plan, own_images, slot_images, recv_scores=None, logits_out=None, labels_out=None, root=0):
        """accel_gather_scores: the score maps plan `plan` has just left in `scores`, from every rank to the root, expanded there into
        logits + labels (device pointers; None on peers)"""
        vp = lambda p: ctypes.c_void_p(p) if p else None
        check(lib().accel_gather_scores(self.handle, plan.handle, int(own_images), int(slot_images), vp(recv_scores), vp(logits_out), vp(labels_out), int(root)))

    def sync(self):
        check(lib().accel_comm_sync(self.handle))

    def close(self):
        if self.handle:
            lib().accel_comm_destroy(self.handle)
            self.handle = ctypes.c_void_p()


class Plan(object):
    def __init__(self, model, handle, role):
        self.model, self.handle, self.role = model, handle, role

    def finalize(self):
        check(lib().accel_plan_finalize(self.handle))

    def run(self):
        check(lib().accel_plan_run(self.handle))

    def ops(self):
        n = lib().accel_plan_num_ops(self.handle)
        out = []
        kind = ctypes.create_string_buffer(32)
        name = ctypes.create_string_buffer(64)
        fl, by = ctypes.c_double(), ctypes.c_double()
        for i in range(n):
            check(lib().accel_plan_op_info(self.handle, i, kind, name, ctypes.byref(fl), ctypes.byref(by)))
            t, k, nw = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
            check(lib().accel_plan_op_launch(self.handle, i, ctypes.byref(t), ctypes.byref(k), ctypes.byref(nw)))
            md = ctypes.c_int()
            check(lib().accel_plan_op_mode(self.handle, i, ctypes.byref(md)))
            out.append({"kind": kind.value.decode(), "name": name.value.decode(), "flops": fl.value, "bytes": by.value,
                        "tile": t.value, "ksplit": k.value, "narrow": nw.value, "mode": md.value})
        return out

    def ranges(self):
        """fp16x2 form: {op name: (pixel scale of the last run, source)} of the convolutions that have one (accel_plan_op_range);
        source 0 = no run yet / all-zero input, 1 = raised by the writers' epilogues, 2 = measured by a pass over the input view"""
        out = {}
        kind = ctypes.create_string_buffer(32)
        name = ctypes.create_string_buffer(64)
        for i in range(lib().accel_plan_num_ops(self.handle)):
            s, src = ctypes.c_float(), ctypes.c_int()
            check(lib().accel_plan_op_range(self.handle, i, ctypes.byref(s), ctypes.byref(src)))
            if s.value:
                check(lib().accel_plan_op_info(self.handle, i, kind, name, None, None))
                out[name.value.decode()] = (s.value, int(src.value))
        return out

    def expand_scores(self, scores_ptr, n_images, logits_ptr, labels_ptr, comm=None):
        """accel_expand_scores: logits + labels of n_images score maps (device pointers) by this plan's own last launch"""
        check(lib().accel_expand_scores(self.handle, ctypes.c_void_p(scores_ptr), int(n_images), ctypes.c_void_p(logits_ptr), ctypes.c_void_p(labels_ptr),
                                        comm.handle if comm is not None else None))

    def run_serial(self):
        """diagnostics: every op in list order on the context stream (no graph replay), then a host wait"""
        check(lib().accel_plan_run_serial(self.handle))

    def arena(self):
        """diagnostics: host copy of the plan's activation arena (uint8)"""
        n = ctypes.c_size_t()
        check(lib().accel_plan_arena_read(self.handle, 0, None, 0, ctypes.byref(n)))
        out = np.empty(n.value, np.uint8)
        check(lib().accel_plan_arena_read(self.handle, 0, _fp(out), out.nbytes, None))
        return out

    def profile(self, iters=3):
        n = lib().accel_plan_num_ops(self.handle)
        ms = np.zeros(n, np.float32)
        check(lib().accel_plan_profile(self.handle, int(iters), _fp(ms), n))
        return ms


class Model(object):
    def __init__(self, ctx):
        self.ctx = ctx
        self.handle = ctypes.c_void_p()
        check(lib().accel_model_create(ctx.handle, ctypes.byref(self.handle)))
        self.plans = {}

    def set_param(self, name, arr):
        a = _f32(arr)
        shape = (ctypes.c_int64 * a.ndim)(*a.shape)
        check(lib().accel_model_set_param(self.handle, name.encode(), _fp(a), a.ndim, shape))

    def set_params(self, *dicts):
        for d in dicts:
            for k, v in d.items():
                self.set_param(k, v.asnumpy() if hasattr(v, "asnumpy") else v)

    def add_plan(self, role, text):
        h = ctypes.c_void_p()
        check(lib().accel_model_add_plan(self.handle, role.encode(), text.encode(), ctypes.byref(h)))
        p = Plan(self, h, role)
        self.plans[role] = p
        return p

    def write(self, buf, arr):
        a = np.ascontiguousarray(arr)
        self.__dict__.get("_resident", {}).pop(buf, None)     # Predictor's record of which host array is in `buf`
        check(lib().accel_model_write(self.handle, buf.encode(), _fp(a), a.nbytes, 0))

    def write_device(self, buf, dev_ptr, nbytes):
        self.__dict__.get("_resident", {}).pop(buf, None)
        check(lib().accel_model_write(self.handle, buf.encode(), ctypes.c_void_p(dev_ptr), nbytes, 1))

    def bind_device(self, buf, dev_ptr, nbytes):
        """zero-copy input (accel_model_bind_device): the plans read the image input `buf` from the caller's HBM buffer until the
        next write / bind; the caller keeps that buffer alive and unchanged until the runs have completed"""
        self.__dict__.get("_resident", {}).pop(buf, None)
        check(lib().accel_model_bind_device(self.handle, buf.encode(), ctypes.c_void_p(dev_ptr), nbytes))

    def read(self, buf, shape, dtype=np.float32):
        out = np.empty(shape, dtype)
        check(lib().accel_model_read(self.handle, buf.encode(), _fp(out), out.nbytes, 0))
        return out

    def read_device(self, buf, dev_ptr, nbytes):
        """enqueue a D2D copy of a persistent buffer into caller-owned HBM (no host sync)"""
        check(lib().accel_model_read(self.handle, buf.encode(), ctypes.c_void_p(dev_ptr), nbytes, 1))

    def has_buffer(self, buf):
        ptr, n = ctypes.c_void_p(), ctypes.c_size_t()
        return lib().accel_model_buffer(self.handle, buf.encode(), ctypes.byref(ptr), ctypes.byref(n)) == 0

    def buffer(self, buf):
        ptr, n = ctypes.c_void_p(), ctypes.c_size_t()
        check(lib().accel_model_buffer(self.handle, buf.encode(), ctypes.byref(ptr), ctypes.byref(n)))
        return ptr.value, n.value

    def generation(self, buf):
        """write generation of a persistent buffer (accel_model_buffer_generation)"""
        g = ctypes.c_uint64()
        check(lib().accel_model_buffer_generation(self.handle, buf.encode(), ctypes.byref(g)))
        return int(g.value)

    def prefetch(self, buf, pinned):
        """enqueue the upload of a PinnedBuffer into the shadow of `buf` on the copy stream (overlaps the running plan)"""
        check(lib().accel_model_prefetch(self.handle, buf.encode(), pinned.ptr, pinned.nbytes))

    def commit(self, buf):
        self.__dict__.get("_resident", {}).pop(buf, None)
        check(lib().accel_model_commit(self.handle, buf.encode()))

    def read_async(self, buf, pinned):
        """enqueue the download of `buf` into a PinnedBuffer on the compute stream; valid after ctx.sync()"""
        check(lib().accel_model_read_async(self.handle, buf.encode(), pinned.ptr, pinned.nbytes))

    def _outs(self, want, H, W, ncls=19):
        feat = np.empty((1, 2048, H // 16, W // 16), np.float32) if "feat" in want else None
        logits = np.empty((1, ncls, H, W), np.float32) if "logits" in want else None
        labels = np.empty((1, H, W), np.uint8) if "labels" in want else None
        return feat, logits, labels

    def key_forward(self, img, want=("logits", "labels")):
        """accel_key_forward: host image in, requested host outputs back (dict)."""
        img = _f32(img)
        H, W = img.shape[-2:]
        f, lg, lb = self._outs(want, H, W)
        self.__dict__.get("_resident", {}).clear()
        check(lib().accel_key_forward(self.handle, _fp(img), 0, None if f is None else _fp(f), None if lg is None else _fp(lg),
                                      None if lb is None else _fp(lb), 0))
        return {k: v for k, v in (("feat", f), ("logits", lg), ("labels", lb)) if v is not None}

    def cur_forward(self, img_cur, img_prev, want=("logits", "labels")):
        a, b = _f32(img_cur), _f32(img_prev)
        H, W = a.shape[-2:]
        f, lg, lb = self._outs(want, H, W)
        self.__dict__.get("_resident", {}).clear()
        check(lib().accel_cur_forward(self.handle, _fp(a), _fp(b), 0, None if f is None else _fp(f),
                                      None if lg is None else _fp(lg), None if lb is None else _fp(lb), 0))
        return {k: v for k, v in (("feat", f), ("logits", lg), ("labels", lb)) if v is not None}

    def close(self):
        if self.handle:
            lib().accel_model_destroy(self.handle)
            self.handle = ctypes.c_void_p()
