"""accel_model_bind_device: the image inputs read from the caller's HBM buffers instead of from copies in the model's own input
buffers (what bench.py does for its resident clips).  The frames of a step must come out bit-identical either way, a later
write into the buffer must end the binding, and buffers other than image inputs must be refused."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _logits_of_a_step(wl, monkeypatch, copy):
    import torch
    outs = []
    m = wl.model
    put = m.write_device if copy else m.bind_device
    for t in range(wl.interval):
        put("data", wl.dev_frames[t].data_ptr(), wl.nbytes)
        if t == 0:
            wl.key.run()
        else:
            put("data_key", wl.dev_frames[t - 1].data_ptr(), wl.nbytes)
            (wl.cur if t % 2 else wl.cur_b).run()
        outs.append(m.read("logits", (wl.B, 19, wl.H, wl.W)).copy())
    torch.cuda.synchronize()
    return outs


def test_bound_frames_give_the_same_logits_as_copied_frames(demo_cfg, monkeypatch):
    import bench
    from accel_amd.runtime import AccelError
    H, W = 256, 512
    demo_cfg.SCALES[0] = (H, W)
    wl = bench.Workload("18", 2, H, W, 3, 0, 0, demo_cfg)
    try:
        a = _logits_of_a_step(wl, monkeypatch, copy=True)
        b = _logits_of_a_step(wl, monkeypatch, copy=False)
        c = _logits_of_a_step(wl, monkeypatch, copy=True)       # a write ends the binding
        for t in range(3):
            assert np.array_equal(a[t], b[t]), "frame %d: bound input differs by %g" % (t, float(np.abs(a[t] - b[t]).max()))
            assert np.array_equal(a[t], c[t])
        assert float(np.abs(a[0] - a[1]).max()) > 1e-3, "the frames of the clip must differ for this test to mean anything"
        # the bound buffer is read at run time: new contents in the SAME caller buffer are seen without a new bind
        wl.model.bind_device("data", wl.dev_frames[0].data_ptr(), wl.nbytes)
        wl.key.run()
        k0 = wl.model.read("logits", (wl.B, 19, H, W)).copy()
        keep = wl.dev_frames[0].clone()
        import torch
        torch.cuda.synchronize()
        wl.dev_frames[0].copy_(wl.dev_frames[2])
        torch.cuda.synchronize()                    # torch's copy stream and the library's compute stream are not ordered by themselves
        wl.key.run()
        k2 = wl.model.read("logits", (wl.B, 19, H, W)).copy()
        wl.dev_frames[0].copy_(keep)
        torch.cuda.synchronize()
        assert np.array_equal(k0, a[0]) and not np.array_equal(k2, k0)
        # a read of a bound input returns what the plans read (the caller's frame), not the model-owned copy left by an earlier write
        wl.model.write_device("data", wl.dev_frames[1].data_ptr(), wl.nbytes)
        wl.model.bind_device("data", wl.dev_frames[2].data_ptr(), wl.nbytes)
        assert np.array_equal(wl.model.read("data", (wl.B, 3, H, W)), wl.dev_frames[2].cpu().numpy())
        # a raw-pointer hand-out ends the binding: what the caller then writes through the pointer is what the plans read
        # (the Predictor path does exactly this: tester.py takes m.buffer('data_key') and copies the previous frame into it)
        ptr, nb = wl.model.buffer("data")
        assert nb == wl.nbytes
        torch_dst = __import__("accel_amd.dist", fromlist=["as_torch"]).as_torch(ptr, (wl.B, 3, H, W))
        torch_dst.copy_(wl.dev_frames[0])
        __import__("torch").cuda.synchronize()      # the copy ran on torch's stream, the plan runs on the library's
        wl.key.run()
        assert np.array_equal(wl.model.read("logits", (wl.B, 19, H, W)), a[0]), "a write through accel_model_buffer's pointer was ignored"
        with pytest.raises(AccelError, match="not an image input"):
            wl.model.bind_device("feat", wl.dev_frames[0].data_ptr(), wl.nbytes)
        with pytest.raises(AccelError, match="bytes"):
            wl.model.bind_device("data", wl.dev_frames[0].data_ptr(), wl.nbytes - 4)
    finally:
        wl.model.close() if hasattr(wl.model, "close") else None
