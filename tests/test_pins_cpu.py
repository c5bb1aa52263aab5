"""Pins that do not depend on this repository's own writers / restatements ("pin what can be
pinned independently"):

 * `.params` reader against byte streams assembled BY HAND from the format description
   (lib/utils/save_model.py:4-18 = mx.nd.save of {"arg:..", "aux:.."}; MXNet ndarray.cc NDArray::Save / Load for
   the legacy, V1/V2 and V3 array headers) -- every byte below is written out literally with its meaning, none of it
   comes from accel_amd.utils.load_model.nd_save;
 * the oracle's deformable convolution against a second, structurally different restatement built from
   torch.nn.functional.grid_sample (one bilinear gather per kernel tap and deformable group, contraction in float64)
   for samples that stay inside the image, where the DCN-v1 rule and zero-padded bilinear sampling coincide.
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from accel_amd.utils import load_model as L
from oracle import ops as O


def _hex(*parts):
    return bytes.fromhex("".join(p.split("#")[0] for part in parts for p in part.splitlines()).replace(" ", ""))


# ---- legacy layout (MXNet <= 0.11 = the reference's pinned commit 62ecb60) -------------------------------------
LEGACY = _hex("""
12 01 00 00 00 00 00 00      # uint64 list magic 0x112
00 00 00 00 00 00 00 00      # uint64 reserved
03 00 00 00 00 00 00 00      # uint64 number of arrays = 3
02 00 00 00                  # array 0: uint32 ndim = 2
02 00 00 00  03 00 00 00     #   uint32 dims (2, 3)
01 00 00 00  00 00 00 00     #   context: int32 dev_type = 1 (cpu), int32 dev_id = 0
00 00 00 00                  #   int32 type_flag = 0 (float32)
00 00 80 3F  00 00 00 40  00 00 40 40   # 1.0 2.0 3.0
00 00 80 40  00 00 A0 40  00 00 C0 40   # 4.0 5.0 6.0
01 00 00 00                  # array 1: ndim = 1
03 00 00 00                  #   dims (3,)
02 00 00 00  01 00 00 00     #   context: dev_type = 2 (gpu), dev_id = 1  (saved from a GPU array: must be ignored)
00 00 00 00                  #   float32
00 00 00 3F  00 00 C0 3F  00 00 00 40   # 0.5 1.5 2.0
04 00 00 00                  # array 2: ndim = 4
01 00 00 00  02 00 00 00  01 00 00 00  01 00 00 00   # dims (1, 2, 1, 1)
01 00 00 00  00 00 00 00     #   cpu(0)
00 00 00 00                  #   float32
00 00 80 BF  00 00 80 3E     # -1.0 0.25
03 00 00 00 00 00 00 00      # uint64 number of names = 3
0E 00 00 00 00 00 00 00      # uint64 length 14
""", "61 72 67 3A 66 63 36 5F 77 65 69 67 68 74",                     # "arg:fc6_weight"
                 "11 00 00 00 00 00 00 00", "61 75 78 3A 62 6E 5F 6D 6F 76 69 6E 67 5F 76 61 72",        # 17: "aux:bn_moving_var"
                 "14 00 00 00 00 00 00 00", "61 72 67 3A 63 6F 6E 76 5F 74 65 73 74 5F 77 65 69 67 68 74")  # 20: "arg:conv_test_weight"

# ---- V2 (MXNet 0.12 .. 1.5): per-array magic, storage type, int64 dims; here with float16 / float64 / uint8 payloads ----
V2 = _hex("""
12 01 00 00 00 00 00 00   00 00 00 00 00 00 00 00   03 00 00 00 00 00 00 00
C9 FA 93 F9                  # array 0: uint32 magic 0xF993FAC9 (V2)
00 00 00 00                  #   int32 storage type 0 (dense)
01 00 00 00                  #   uint32 ndim = 1
02 00 00 00 00 00 00 00      #   int64 dims (2,)
01 00 00 00  00 00 00 00     #   cpu(0)
02 00 00 00                  #   type_flag 2 = float16
00 3C  00 C0                 #   1.0, -2.0
C9 FA 93 F9  00 00 00 00  02 00 00 00                       # array 1: V2, dense, ndim 2
01 00 00 00 00 00 00 00  02 00 00 00 00 00 00 00            #   dims (1, 2)
01 00 00 00  00 00 00 00  01 00 00 00                       #   cpu(0), type_flag 1 = float64
00 00 00 00 00 00 F0 3F   00 00 00 00 00 00 04 C0           #   1.0, -2.5
C8 FA 93 F9                  # array 2: magic 0xF993FAC8 (V1: no storage-type field)
01 00 00 00                  #   ndim 1
03 00 00 00 00 00 00 00      #   dims (3,)
01 00 00 00  00 00 00 00  03 00 00 00                       #   cpu(0), type_flag 3 = uint8
07 00 FF                     #   7, 0, 255
03 00 00 00 00 00 00 00
05 00 00 00 00 00 00 00""", "61 72 67 3A 61",                # "arg:a"
           "05 00 00 00 00 00 00 00", "61 72 67 3A 62",      # "arg:b"
           "05 00 00 00 00 00 00 00", "61 75 78 3A 63")      # "aux:c"

# ---- V3 (MXNet >= 1.6, numpy shape semantics): ndim 0 is a SCALAR that carries data ---------------------------------
V3 = _hex("""
12 01 00 00 00 00 00 00   00 00 00 00 00 00 00 00   02 00 00 00 00 00 00 00
CA FA 93 F9  00 00 00 00     # array 0: magic 0xF993FACA (V3), dense
00 00 00 00                  #   int32 ndim = 0: scalar
01 00 00 00  00 00 00 00  00 00 00 00                       #   cpu(0), float32
00 00 20 41                  #   10.0
CA FA 93 F9  00 00 00 00  01 00 00 00                       # array 1: V3, dense, ndim 1
02 00 00 00 00 00 00 00                                     #   dims (2,)
01 00 00 00  00 00 00 00  04 00 00 00                       #   cpu(0), type_flag 4 = int32
FE FF FF FF  2A 00 00 00                                    #   -2, 42
02 00 00 00 00 00 00 00
05 00 00 00 00 00 00 00""", "61 72 67 3A 73",                # "arg:s"
           "05 00 00 00 00 00 00 00", "61 72 67 3A 76")      # "arg:v"


def test_params_legacy_bytes_from_the_format_spec(tmp_path):
    p = tmp_path / "pinned-0007.params"
    p.write_bytes(LEGACY)
    d = L.nd_load(str(p))
    assert list(d) == ["arg:fc6_weight", "aux:bn_moving_var", "arg:conv_test_weight"]
    np.testing.assert_array_equal(d["arg:fc6_weight"], np.array([[1, 2, 3], [4, 5, 6]], np.float32))
    np.testing.assert_array_equal(d["aux:bn_moving_var"], np.array([0.5, 1.5, 2.0], np.float32))
    np.testing.assert_array_equal(d["arg:conv_test_weight"], np.array([-1, 0.25], np.float32).reshape(1, 2, 1, 1))
    assert all(v.dtype == np.float32 for v in d.values())
    # the reference's wrappers on top (lib/utils/load_model.py:4-30,73-93)
    arg, aux = L.load_param(str(tmp_path / "pinned"), 7, process=True)
    assert sorted(arg) == ["conv_weight", "fc6_weight"] and list(aux) == ["bn_moving_var"]
    arg, aux = L.load_checkpoint(str(tmp_path / "pinned"), 7, argprefix="fc6_")
    assert sorted(arg) == ["fc6_conv_test_weight", "fc6_weight"] and list(aux) == ["fc6_bn_moving_var"]


def test_params_v1_v2_bytes_from_the_format_spec(tmp_path):
    p = tmp_path / "v2.params"
    p.write_bytes(V2)
    d = L.nd_load(str(p))
    assert d["arg:a"].dtype == np.float16 and d["arg:a"].tolist() == [1.0, -2.0]
    assert d["arg:b"].dtype == np.float64 and d["arg:b"].shape == (1, 2) and d["arg:b"].tolist() == [[1.0, -2.5]]
    assert d["aux:c"].dtype == np.uint8 and d["aux:c"].tolist() == [7, 0, 255]


def test_params_v3_bytes_from_the_format_spec(tmp_path):
    p = tmp_path / "v3.params"
    p.write_bytes(V3)
    d = L.nd_load(str(p))
    assert d["arg:s"].shape == () and float(d["arg:s"]) == 10.0
    assert d["arg:v"].dtype == np.int32 and d["arg:v"].tolist() == [-2, 42]


def test_params_reader_rejects_damage(tmp_path):
    p = tmp_path / "bad.params"
    p.write_bytes(LEGACY[:-5])                                   # truncated name table
    with pytest.raises(ValueError):
        L.nd_load(str(p))
    p.write_bytes(b"\x13" + LEGACY[1:])                          # wrong list magic
    with pytest.raises(ValueError, match="magic"):
        L.nd_load(str(p))
    sparse = bytearray(V2)
    sparse[28] = 1                                               # storage type of array 0 -> row_sparse
    p.write_bytes(bytes(sparse))
    with pytest.raises(NotImplementedError, match="sparse"):
        L.nd_load(str(p))
    unk = bytearray(V2)
    unk[52] = 9                                                  # type_flag of array 0 -> unknown
    p.write_bytes(bytes(unk))
    with pytest.raises(ValueError, match="type flag"):
        L.nd_load(str(p))


def test_writer_output_is_what_the_spec_says(tmp_path):
    """the writer, checked against the hand-assembled legacy stream (not against the reader)"""
    d = {"arg:fc6_weight": np.array([[1, 2, 3], [4, 5, 6]], np.float32),
         "aux:bn_moving_var": np.array([0.5, 1.5, 2.0], np.float32),
         "arg:conv_test_weight": np.array([-1, 0.25], np.float32).reshape(1, 2, 1, 1)}
    p = tmp_path / "w.params"
    L.nd_save(str(p), d, legacy=True)
    got = bytearray(p.read_bytes())
    want = bytearray(LEGACY)
    want[80:88] = b"\x01\x00\x00\x00\x00\x00\x00\x00"           # the hand stream marks array 1 as gpu(1); the writer says cpu(0)
    assert bytes(got) == bytes(want)


# ---------------------------------------------------------------------------------------------------------------
# deformable convolution: oracle vs a grid_sample-based restatement (interior samples)
# ---------------------------------------------------------------------------------------------------------------
def _dcn_by_grid_sample(x, offset, w, stride, pad, dilate, dg):
    """y[n,k,oy,ox] = sum_{c,i,j} w[k,c,i,j] * bilinear(x[n,c], oy*s - p + i*d + dy, ox*s - p + j*d + dx)
    with (dy, dx) = offset channels (2*(i*kw+j), 2*(i*kw+j)+1) of the deformable group of channel c."""
    N, C, H, W = x.shape
    K, _, kh, kw = w.shape
    Ho = (H + 2 * pad - dilate * (kh - 1) - 1) // stride + 1
    Wo = (W + 2 * pad - dilate * (kw - 1) - 1) // stride + 1
    xt = torch.from_numpy(x).double()
    oy = torch.arange(Ho, dtype=torch.float64)[:, None] * stride - pad
    ox = torch.arange(Wo, dtype=torch.float64)[None, :] * stride - pad
    cpg = C // dg
    cols = torch.zeros(N, C, kh, kw, Ho, Wo, dtype=torch.float64)
    off = torch.from_numpy(offset).double().reshape(N, dg, kh * kw, 2, Ho, Wo)
    for g in range(dg):
        for i in range(kh):
            for j in range(kw):
                py = oy + i * dilate + off[:, g, i * kw + j, 0]
                px = ox + j * dilate + off[:, g, i * kw + j, 1]
                grid = torch.stack([px / ((W - 1) / 2.0) - 1.0, py / ((H - 1) / 2.0) - 1.0], dim=-1)
                s = F.grid_sample(xt[:, g * cpg:(g + 1) * cpg], grid, mode="bilinear", padding_mode="zeros", align_corners=True)
                cols[:, g * cpg:(g + 1) * cpg, i, j] = s
    return torch.einsum("kcij,ncijyx->nkyx", torch.from_numpy(w).double(), cols).numpy()


@pytest.mark.parametrize("dg,stride,pad,dilate", [(1, 1, 2, 2), (4, 1, 2, 2), (2, 2, 1, 1), (1, 1, 0, 1)])
def test_dcn_oracle_vs_grid_sample_interior(dg, stride, pad, dilate):
    rng = np.random.default_rng(100 + dg + 10 * stride)
    N, C, H, W, K, k = 2, 8, 13, 17, 6, 3
    x = rng.standard_normal((N, C, H, W)).astype(np.float32)
    w = rng.standard_normal((K, C, k, k)).astype(np.float32)
    Ho = (H + 2 * pad - dilate * (k - 1) - 1) // stride + 1
    Wo = (W + 2 * pad - dilate * (k - 1) - 1) // stride + 1
    off = rng.uniform(-2.5, 2.5, (N, dg, k * k, 2, Ho, Wo))
    # pull every sample inside [0, H-1] x [0, W-1]: there the DCN-v1 rule (zero outside, clamped high neighbour) and
    # zero-padded bilinear interpolation are the same function
    oy = np.arange(Ho)[:, None] * stride - pad
    ox = np.arange(Wo)[None, :] * stride - pad
    for i in range(k):
        for j in range(k):
            t = i * k + j
            py = oy + i * dilate + off[:, :, t, 0]
            px = ox + j * dilate + off[:, :, t, 1]
            off[:, :, t, 0] += np.clip(py, 0.0, H - 1.0) - py
            off[:, :, t, 1] += np.clip(px, 0.0, W - 1.0) - px
    off = off.reshape(N, dg * 2 * k * k, Ho, Wo).astype(np.float32)
    got = O.deform_conv2d(x, off, w, stride, pad, dilate, dg)
    ref = _dcn_by_grid_sample(x, off, w, stride, pad, dilate, dg)
    assert float(np.abs(got - ref).max()) <= 2e-5 * max(1.0, float(np.abs(ref).max()))


def test_dcn_oracle_vs_grid_sample_also_outside_where_rules_agree():
    """Samples more than one pixel outside the image are zero under both rules; only the one-pixel rim differs
    (DCN-v1: zero for coordinates < 0, clamped high neighbour in [H-1, H)).  Offsets that throw taps far outside
    must therefore still agree."""
    rng = np.random.default_rng(7)
    N, C, H, W, K, k, dg = 1, 4, 9, 11, 3, 3, 1
    x = rng.standard_normal((N, C, H, W)).astype(np.float32)
    w = rng.standard_normal((K, C, k, k)).astype(np.float32)
    off = np.zeros((N, dg, k * k, 2, H, W))
    off[:, :, 0] = -40.0          # tap (0,0): far above / left of the image
    off[:, :, 8] = 55.0           # tap (2,2): far below / right
    off[:, :, 4, 0] = 0.0         # centre tap stays on the pixel
    off = off.reshape(N, dg * 2 * k * k, H, W).astype(np.float32)
    got = O.deform_conv2d(x, off, w, 1, 1, 1, dg)
    ref = _dcn_by_grid_sample(x, off, w, 1, 1, 1, dg)
    # taps 1..7 except the centre sit on integer positions; the rim rows/cols of taps hitting -1 or H are zero in both
    assert float(np.abs(got - ref).max()) <= 2e-5 * max(1.0, float(np.abs(ref).max()))


# ---------------------------------------------------------------------------------------------------------------
# lib/utils/image.py:194-222 resize = cv2.resize(fx, fy, INTER_LINEAR): hand-computed cases for non-identity scales
# ---------------------------------------------------------------------------------------------------------------
def test_resize_inter_linear_hand_computed_cases():
    from accel_amd.utils.image import resize
    # x2: source coordinate (d + 0.5) / 2 - 0.5 = -0.25, 0.25, 0.75, 1.25 -> clamped 0 | 0.25 | 0.75 | 1
    im = np.array([[[0.0], [100.0]], [[200.0], [255.0]]])
    out, sc = resize(im, 4, 4)
    assert sc == 2.0 and out.shape == (4, 4, 1)
    np.testing.assert_allclose(out[0, :, 0], [0.0, 25.0, 75.0, 100.0])
    np.testing.assert_allclose(out[3, :, 0], [200.0, 213.75, 241.25, 255.0])
    np.testing.assert_allclose(out[1, :, 0], 0.75 * out[0, :, 0] + 0.25 * out[3, :, 0])
    # x0.5: (d + 0.5) * 2 - 0.5 = 0.5, 2.5 -> the mean of pixels (0,1) and (2,3) in each direction
    im4 = np.arange(16, dtype=np.float64).reshape(4, 4, 1)
    half, sc = resize(im4, 2, 2)
    assert sc == 0.5
    np.testing.assert_allclose(half[:, :, 0], [[2.5, 4.5], [10.5, 12.5]])
    # fx = 0.7 on 10 columns: dsize = round(7.0) = 7 and the source step is 1 / 0.7 (the requested factor); on 3 columns
    # dsize = round(2.1) = 2 while in / out = 1.5: cv2 keeps 1 / 0.7, so d = 0, 1 read x = 0.2143, 1.6429
    row = np.array([[[0.0], [10.0], [40.0]]])
    r, sc = resize(np.repeat(row, 3, axis=0), 2.1, 2.1)      # target / short side = 0.7
    assert abs(sc - 0.7) < 1e-12 and r.shape == (2, 2, 1)
    x = (np.arange(2) + 0.5) / 0.7 - 0.5
    exp = [0.0 * (1 - x[0]) + 10.0 * x[0], 10.0 * (2 - x[1]) + 40.0 * (x[1] - 1)]
    np.testing.assert_allclose(r[0, :, 0], exp)
    # uint8 stays uint8, rounded once
    u8, _ = resize(np.array([[[0], [101]], [[0], [101]]], np.uint8), 4, 4)
    assert u8.dtype == np.uint8 and u8[0, :, 0].tolist() == [0, 25, 76, 101]
    # stride padding: zeros to the next multiple (image.py:213-220)
    pad, _ = resize(np.ones((5, 7, 3)), 5, 7, stride=4)
    assert pad.shape == (8, 8, 3) and pad[:5, :7].min() == 1 and pad[5:].max() == 0 and pad[:, 7:].max() == 0


# ---- the published examples of the dependency itself ---------------------------------------------------------------------------
# MXNet is un-vendored (the path's arithmetic lives in mxnet @ 62ecb60), but its operator documentation carries worked
# examples; the two below are the docstring of `BilinearSampler` (src/operator/bilinear_sampler.cc, "Example 1" zoom out by an affine
# grid, "Example 2" the `warp` grid this path uses: accel_18.py:174-175, resnet_v1_101_flownet_deeplab.py get_*_test_symbol).
DOC_DATA = np.array([[[[1, 4, 3, 6], [1, 8, 8, 9], [0, 4, 1, 5], [1, 0, 1, 3]]]], np.float32)


def test_mxnet_docstring_example_warp_grid_and_bilinear_sampler():
    """flow = (1, 0) everywhere: GridGenerator(transform_type='warp') + BilinearSampler shift the image one pixel to the left,
    the column that would come from outside is zero"""
    flow = np.zeros((1, 2, 4, 4), np.float32)
    flow[:, 0] = 1.0                                       # channel 0 = horizontal displacement
    want = np.array([[[[4, 3, 6, 0], [8, 8, 9, 0], [4, 1, 5, 0], [0, 1, 3, 0]]]], np.float32)
    got = O.bilinear_sampler(DOC_DATA, O.grid_generator_warp(flow))
    assert np.array_equal(got, want)
    assert np.array_equal(O.flow_warp(DOC_DATA, flow), want)
    # the grid itself, from the operator's definition: x_src = (x + flow_x) / ((W - 1) / 2) - 1
    grid = O.grid_generator_warp(flow)
    xs = (np.arange(4, dtype=np.float32) + 1) / np.float32(1.5) - 1
    assert np.allclose(grid[0, 0], np.tile(xs, (4, 1)), atol=1e-6)
    assert np.allclose(grid[0, 1], np.tile(((np.arange(4, dtype=np.float32)) / np.float32(1.5) - 1)[:, None], (1, 4)), atol=1e-6)


def test_mxnet_docstring_example_affine_zoom_out_through_bilinear_sampler():
    """affine_matrix = [[2, 0, 0], [0, 2, 0]], target 4x4: the sampler sees the grid 2 * (x_t, y_t) with x_t, y_t in {-1, -1/3, 1/3, 1};
    the documentation's output has the four interior means and zeros where the grid leaves [-1, 1]"""
    t = np.linspace(-1, 1, 4, dtype=np.float32)
    grid = np.stack([np.tile(2 * t, (4, 1)), np.tile((2 * t)[:, None], (1, 4))])[None].astype(np.float32)
    want = np.array([[[[0, 0, 0, 0], [0, 3.5, 6.5, 0], [0, 1.25, 2.5, 0], [0, 0, 0, 0]]]], np.float32)
    assert np.allclose(O.bilinear_sampler(DOC_DATA, grid), want, atol=1e-6)


def test_mxnet_documented_output_size_rules():
    """Pooling: 'valid' floor((x + 2p - k) / s) + 1, 'full' ceil((x + 2p - k) / s) + 1 (operator docs); Deconvolution:
    (x - 1) s - 2p + k.  The sizes the path depends on: pool1 of the ResNet-101 stem ('full': 512 -> 256, 514 -> 257 where 'valid'
    gives 256), pooling0 of the ResNet-18 branch (pad 1: 512 -> 256), FlowNet's 4x4/2 deconvolutions (pad 1: x -> 2x; pad 0 then Crop:
    x -> 2x + 2), the 32x32/16 score upsampler (x -> 16 x + 16, cropped by 8)"""
    size = lambda n, k, s, p, conv: O.pool2d(np.zeros((1, 1, n, n), np.float32), "max", k, s, p, conv).shape[-1]
    assert size(512, 3, 2, 0, "full") == 256 and size(514, 3, 2, 0, "full") == 257 and size(514, 3, 2, 0, "valid") == 256
    assert size(512, 3, 2, 1, "valid") == 256 and size(7, 3, 2, 0, "full") == 3 and size(8, 3, 2, 0, "full") == 4 and size(8, 3, 2, 0, "valid") == 3
    dec = lambda n, k, s, p: O.deconv2d(np.zeros((1, 1, n, n), np.float32), np.zeros((1, 1, k, k), np.float32), None, s, p).shape[-1]
    assert dec(16, 4, 2, 1) == 32 and dec(16, 4, 2, 0) == 34 and dec(64, 32, 16, 0) == 16 * 64 + 16
