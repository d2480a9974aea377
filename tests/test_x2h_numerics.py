"""CPU statement of the x2h operand arithmetic (csrc/gemm.hip.h: x2h_split8 / x2h_dyn_scale): the bounds the
documentation quotes, checked with numpy's IEEE fp16 (round to nearest even, the rounding of v_cvt_pk_f16_f32).  The device
kernels are checked against the oracle in tests/test_gpu_parity.py; this file pins the arithmetic they are built on."""
import numpy as np


def split(xs):
    """hi = fp16(xs), lo = fp16(xs - hi) for fp32 xs (already scaled)."""
    xs = xs.astype(np.float32)
    hi = xs.astype(np.float16)
    lo = (xs - hi.astype(np.float32)).astype(np.float16)
    return hi, lo


def dyn_scale(m):
    """x2h_dyn_scale: (s, 1/s) from the float bits of a maximum."""
    bits = np.float32(m).view(np.uint32)
    ef = int(bits >> 23)
    ef = 24 if ef < 24 else (250 if ef > 250 else ef)
    s = np.uint32((267 - ef) << 23).view(np.float32)
    inv = np.uint32((ef - 13) << 23).view(np.float32)
    return s, inv


def test_two_plane_representation_error():
    rng = np.random.default_rng(0)
    x = (rng.standard_normal(200000) * np.exp(rng.uniform(-2.0, 6.0, 200000))).astype(np.float32)
    x = x[np.abs(x) >= 2.0 ** -3]          # the normal range of the low plane
    x = x[np.abs(x) < 65000.0]
    hi, lo = split(x)
    err = np.abs(hi.astype(np.float64) + lo.astype(np.float64) - x.astype(np.float64))
    rel = err / np.abs(x.astype(np.float64))
    assert rel.max() <= 2.0 ** -22
    assert np.sqrt((rel ** 2).mean()) < 2.0 ** -23            # rms 2^-23.6
    # below the normal range of the low plane the error is absolute: half a subnormal step
    small = (rng.uniform(-1, 1, 100000) * 2.0 ** -3).astype(np.float32)
    hi, lo = split(small)
    assert np.abs(hi.astype(np.float64) + lo.astype(np.float64) - small.astype(np.float64)).max() <= 2.0 ** -25
    # out of range is loud
    with np.errstate(over="ignore"):
        hi, lo = split(np.array([70000.0], np.float32))
    assert np.isinf(hi[0])


def test_three_term_product_against_the_exact_one():
    rng = np.random.default_rng(1)
    a = rng.uniform(0.0, 8.0, (64, 256)).astype(np.float32) * 64.0          # activations x 2^6
    w = (rng.standard_normal((256, 256)) * 0.09).astype(np.float32) * 256.0    # weights x 2^8
    ah, al = (p.astype(np.float64) for p in split(a))
    wh, wl = (p.astype(np.float64) for p in split(w))
    got = (ah @ wh.T + ah @ wl.T + al @ wh.T) / (64.0 * 256.0)      # the three kept terms (exact accumulation here)
    ref = (a.astype(np.float64) / 64.0) @ (w.astype(np.float64) / 256.0).T
    mag = (np.abs(a).astype(np.float64) / 64.0) @ (np.abs(w).astype(np.float64) / 256.0).T
    # every product is off by at most 2^-22 (a) + 2^-22 (w) + 2^-22 (dropped lo x lo) of its magnitude; independent signs
    assert np.abs(got - ref).max() <= 3 * 2.0 ** -22 * mag.max()
    assert np.sqrt(((got - ref) ** 2).mean()) <= 2.0 ** -24 * np.sqrt((mag ** 2).mean())   # far below fp32's own accumulation noise


def test_dynamic_scale_is_an_exact_power_of_two_and_places_the_maximum():
    for m in (1e-30, 3.7e-9, 2.5e-4, 1.0, 1.999, 2.0, 123456.0, 9.9e8, 1e30):
        s, inv = dyn_scale(m)
        assert s * inv == 1.0
        assert np.log2(float(s)) == round(np.log2(float(s)))
        assert 2.0 ** 13 <= float(np.float32(m)) * float(s) < 2.0 ** 14
    # zero maximum (all-zero adjoint) and the clamps: finite normal factors
    for m in (0.0, 1e-38, 3e38):
        s, inv = dyn_scale(m)
        assert np.isfinite(s) and np.isfinite(inv) and s > 0 and inv > 0 and s * inv == 1.0
    # scaling the data by 2^k scales the recorded maximum by 2^k and the scale by 2^-k: scaled operands are bit-identical
    x = np.float32(0.0371)
    for k in (-40, 17, 40):
        m0, m1 = np.float32(0.9), np.float32(0.9) * np.float32(2.0) ** k
        s0, _ = dyn_scale(m0)
        s1, _ = dyn_scale(m1)
        assert x * s0 == (x * np.float32(2.0) ** k) * s1
