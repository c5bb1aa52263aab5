"""The arithmetic behind the bf16x3 launch geometries (accel_amd/csrc/conv_igemm.hip, conv_igemm_b3_kernel), restated in
numpy: the three-way split of an fp32 value into bf16 terms is EXACT, and the six products the kernel keeps reproduce
the fp32 product to better than one fp32 rounding.  (The GPU side of the claim is tests/test_bf16x3_gpu.py.)"""
import numpy as np


def split3(x):
    """top 16 bits of x, of the residual, of its residual -- what split3_pair() / pack_bf16x3() do"""
    x = np.asarray(x, np.float32)
    terms, r = [], x.copy()
    for _ in range(3):
        t = (r.view(np.uint32) & np.uint32(0xFFFF0000)).view(np.float32)
        terms.append(t)
        r = (r - t).astype(np.float32)          # exact: r and t share sign and exponent range
    return terms, r


def _values(n, seed):
    rng = np.random.default_rng(seed)
    v = rng.standard_normal(n).astype(np.float32) * np.float32(2.0) ** rng.integers(-20, 20, n).astype(np.float32)
    return np.concatenate([v, np.float32([0.0, -0.0, 1.0, -1.0, 3.0e38, -3.0e38, 1.0e-30, 255.0, 1.0 / 3.0])])


def test_three_way_split_is_exact():
    """(for |x| >= 2^-110: below that the residual terms are subnormal and keep fewer bits -- 1e-33 of the activations' scale)"""
    x = _values(200000, 1)
    (a0, a1, a2), rest = split3(x)
    for t in (a0, a1, a2):
        assert not np.any(t.view(np.uint32) & np.uint32(0xFFFF)), "a term is not representable in bf16"
    # the three terms sum to the value bit for bit (in any precision >= fp32 that holds 24 significant bits: float64 here)
    assert np.array_equal((a0.astype(np.float64) + a1.astype(np.float64) + a2.astype(np.float64)).astype(np.float32), x)
    assert np.array_equal(a0.astype(np.float64) + a1.astype(np.float64) + a2.astype(np.float64), x.astype(np.float64))
    assert not np.any(rest), "24 significant bits fit three 8-bit terms: nothing may be left"
    # magnitudes: each term is at most 2^-8 of the previous one's binade
    nz = a0 != 0
    assert np.all(np.abs(a1[nz]) <= np.abs(a0[nz]) * 2.0 ** -7) and np.all(np.abs(a2[nz]) <= np.abs(a0[nz]) * 2.0 ** -15)


def test_six_products_reproduce_the_fp32_product():
    a, b = _values(100000, 2)[:100000], _values(100000, 3)[:100000]
    (a0, a1, a2), _ = split3(a)
    (b0, b1, b2), _ = split3(b)
    f = lambda t: t.astype(np.float64)
    kept = f(a1) * f(b1) + f(a0) * f(b2) + f(a2) * f(b0) + f(a0) * f(b1) + f(a1) * f(b0) + f(a0) * f(b0)
    exact = f(a) * f(b)
    scale = np.abs(exact)
    ok = scale > 0
    rel = np.abs(kept - exact)[ok] / scale[ok]
    # dropped: a1*b2 + a2*b1 + a2*b2 <= 2 * 2^-7 * 2^-15 + 2^-30 of |a0*b0|: under 2^-21 relative, typically 2^-24
    assert float(rel.max()) <= 2.0 ** -21
    assert float(np.median(rel)) <= 2.0 ** -24
    # every kept term is exact in fp32 (8 x 8 significant bits), so the MFMA's fp32 accumulation is the only rounding left
    for p, q in ((a1, b1), (a0, b2), (a2, b0), (a0, b1), (a1, b0), (a0, b0)):
        prod = f(p) * f(q)
        fin = np.isfinite(prod) & (np.abs(prod) < 3.0e38) & ((np.abs(prod) > 1.2e-38) | (prod == 0))
        assert np.array_equal(prod[fin].astype(np.float32).astype(np.float64), prod[fin])


def test_long_dot_products_stay_at_fp32_accumulation_noise():
    rng = np.random.default_rng(4)
    K = 4608
    a = rng.standard_normal((64, K)).astype(np.float32) * 3
    b = (rng.standard_normal((K, 32)) * (2.0 / K) ** 0.5).astype(np.float32)
    (a0, a1, a2), _ = split3(a)
    (b0, b1, b2), _ = split3(b)
    acc = np.zeros((64, 32), np.float32)
    for p, q in ((a1, b1), (a0, b2), (a2, b0), (a0, b1), (a1, b0), (a0, b0)):      # fp32 accumulation, as the MFMA does
        acc = (acc + (p.astype(np.float64) @ q.astype(np.float64)).astype(np.float32)).astype(np.float32)
    truth = a.astype(np.float64) @ b.astype(np.float64)
    plain = (a @ b).astype(np.float64)                                              # numpy's own fp32 GEMM
    scale = np.abs(truth).max()
    e3, e32 = np.abs(acc - truth).max() / scale, np.abs(plain - truth).max() / scale
    assert e3 <= 5e-7 and e3 <= 4 * e32 + 2e-7, (e3, e32)
