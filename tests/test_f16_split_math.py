"""The arithmetic behind the default fp32-grade kernels (csrc/field_eval_split16_impl.h, MVS16_F16 = 1; csrc/train_ops.hip, MVT_BWD_F16),
restated in NumPy and checked on the CPU: an fp32 operand as two fp16 pieces, a product as three piece products.
  weights      tw = 64 w :  A0 = rn16(tw),  A1 = rn16(tw - A0),        A0s = A0 / 64
  activations  tv = v/64 :  B0 = rn16(tv),  B1 = rn16(v - 64 B0) = rn16(64 (tv - B0))
  product      A0s B1 + A1 B0 + A0 B0  ~  w v
The GPU tests (tests/test_gpu_split.py) measure the kernels themselves; this file pins the claims made about the representation."""
import numpy as np

F32, F16, F64 = np.float32, np.float16, np.float64


def split_weight(w):
    tw = (w * F32(64)).astype(F32)
    a0 = tw.astype(F16)
    r = (tw - a0.astype(F32)).astype(F32)
    a1 = r.astype(F16)
    a0s = (a0.astype(F32) * F32(1 / 64)).astype(F16)
    return a0, a0s, a1, r, tw


def split_activation(v):
    tv = (v * F32(1 / 64)).astype(F32)
    b0 = tv.astype(F16)
    r64 = (v - F32(64) * b0.astype(F32)).astype(F32)           # what v_fma_mix_f32(B0, -64, v) returns
    b1 = r64.astype(F16)
    return b0, b1, r64, tv


def test_remainders_are_exact_in_fp32():
    rng = np.random.default_rng(0)
    w = (rng.standard_normal(200000) * 0.1).astype(F32)
    v = np.abs(rng.standard_normal(200000) * 0.7).astype(F32)
    a0, _, _, r, tw = split_weight(w)
    assert np.array_equal(r.astype(F64), tw.astype(F64) - a0.astype(F64))
    b0, _, r64, _ = split_activation(v)
    assert np.array_equal(r64.astype(F64), v.astype(F64) - 64.0 * b0.astype(F64))


def test_two_pieces_represent_an_operand_to_23_bits():
    rng = np.random.default_rng(1)
    w = (rng.standard_normal(400000) * 0.1).astype(F32)
    a0, a0s, a1, _, _ = split_weight(w)
    rec = (a0.astype(F64) + a1.astype(F64)) / 64.0
    assert (np.abs(rec - w) <= 2.0 ** -23 * np.abs(w) + 2.0 ** -31).all()          # 2^-31 = 2^-25 / 64: where the unscaled remainder is subnormal
    big = np.abs(w) >= 2.0 ** -8                                                    # 64 |w| >= 1/4: every remainder piece keeps 11 bits
    assert (np.abs(rec - w)[big] <= 2.0 ** -23 * np.abs(w)[big]).all()
    normal = np.abs(a0.astype(F64)) >= 2.0 ** -8
    assert np.array_equal(a0s.astype(F64)[normal], a0.astype(F64)[normal] / 64.0)   # A0 / 64 is exact while it stays normal
    v = np.abs(rng.standard_normal(400000) * 0.7).astype(F32)
    b0, b1, _, _ = split_activation(v)
    rec = 64.0 * b0.astype(F64) + b1.astype(F64)
    assert (np.abs(rec - v) <= 2.0 ** -23 * np.abs(v) + 2.0 ** -25).all()           # 2^-25: the scaled remainder 64 (tv - B0) is subnormal
    big = np.abs(v) >= 2.0 ** -2
    assert (np.abs(rec - v)[big] <= 2.0 ** -23 * np.abs(v)[big]).all()


def test_three_piece_products_against_the_exact_product():
    rng = np.random.default_rng(2)
    w = (rng.standard_normal(400000) * 0.1).astype(F32)
    v = np.abs(rng.standard_normal(400000) * 0.7).astype(F32)
    a0, a0s, a1, _, _ = split_weight(w)
    b0, b1, _, _ = split_activation(v)
    f = lambda x: x.astype(F64)
    got = f(a0s) * f(b1) + f(a1) * f(b0) + f(a0) * f(b0)                            # products of fp16 pairs are exact in the MFMA
    exact = f(w) * f(v)
    err = np.abs(got - exact)
    # worst case: representation 2 x 2^-23 + dropped term 2^-22, plus the subnormal floors of the two remainders
    assert (err <= 2.0 ** -21 * np.abs(exact) + 2.0 ** -24 * np.abs(f(w)) + 2.0 ** -30 * np.abs(f(v))).all()
    full = (np.abs(w) >= 2.0 ** -8) & (np.abs(v) >= 0.25)                           # both operands above their subnormal floors
    rel = err[full] / np.abs(exact)[full]
    assert rel.max() <= 2.0 ** -21 and np.sqrt(np.mean(rel ** 2)) < 2.0 ** -23      # rms at fp32 rounding level
    # a 128-term dot product: its error is below the fp32 accumulation noise of the same sum
    W, V = w[:128 * 2000].reshape(2000, 128), v[:128 * 2000].reshape(2000, 128)
    G = got[:128 * 2000].reshape(2000, 128).sum(1)
    E = (f(W) * f(V)).sum(1)
    scale = (np.abs(f(W)) * np.abs(f(V))).sum(1)
    assert (np.abs(G - E) <= 2.0 ** -23 * scale).all()


def test_gradient_scale_keeps_the_pieces_normal():
    """train_ops.hip amax_scale: e = 140 - biased exponent of max |g|, i.e. max |g| 2^e in [2^13, 2^14); pieces of g 2^e / 64."""
    rng = np.random.default_rng(3)
    for mag in (1e-9, 3e-6, 2e-2, 7.0):
        g = (rng.standard_normal(100000) * mag).astype(F32)
        m = np.abs(g).max()
        ex = (np.array([m], F32).view(np.int32)[0] >> 23) & 0xff
        e = 140 - int(ex)
        sc = F32(2.0 ** e)
        assert 2.0 ** 13 <= float(m) * float(sc) < 2.0 ** 14
        u = (g * sc).astype(F32)
        b0 = (u * F32(1 / 64)).astype(F32).astype(F16)
        b1 = (u - F32(64) * b0.astype(F32)).astype(F32).astype(F16)
        assert np.isfinite(b0.astype(F32)).all() and np.isfinite(b1.astype(F32)).all()
        rec = (64.0 * b0.astype(F64) + b1.astype(F64)) / float(sc)
        big = np.abs(g) >= float(m) * 2.0 ** -15
        assert (np.abs(rec - g)[big] <= 2.0 ** -23 * np.abs(g)[big]).all()
        assert (np.abs(rec - g) <= 2.0 ** -23 * np.abs(g) + 2.0 ** -38 * float(m)).all()
