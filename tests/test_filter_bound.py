"""CPU test of the tile kernel's Gram-form filter bound (csrc/kernels_nonbonded.hip.hpp, phase 1): the f32 evaluation of
|r|^2 - 2 r.c < (cut2 + 1e-4) - |c|^2 must keep every pair whose squared distance, evaluated exactly from the same f32 inputs,
is below cut2 -- for every |row component| + |column component| below TM_GRAM_MAX_EXTENT = 4 nm, the range the kernel uses the
form for.  The device arithmetic (one fma per term, f32) is emulated in numpy: products of f32 values are exact in f64, so a
f64 fma rounded to f32 differs from the device's single rounding only in astronomically rare double-rounding cases."""
import numpy as np


def f32(x):
    return np.asarray(x, dtype=np.float32)


def fma32(a, b, c):
    return f32(f32(a).astype(np.float64) * f32(b).astype(np.float64) + f32(c).astype(np.float64))


def gram_accepts(r, c, cut2, with_w):
    """the kernel's arithmetic: rq = fma chain over the row's components, cc likewise, thr = (cut2 + 1e-4) - cc,
    acc = rq + sum r_i * (-2 c_i) as fma steps; accept iff acc < thr"""
    n = 4 if with_w else 3
    rq = f32(r[:, 0] * r[:, 0])
    cc = f32(c[:, 0] * c[:, 0])
    for i in range(1, n):
        rq = fma32(r[:, i], r[:, i], rq)
        cc = fma32(c[:, i], c[:, i], cc)
    thr = f32(f32(np.float32(cut2) + np.float32(1e-4)) - cc)
    acc = rq
    for i in range(n):
        acc = fma32(r[:, i], f32(-2.0) * c[:, i], acc)
    return acc < thr


def test_gram_filter_never_drops_a_pair_inside_the_cutoff():
    rng = np.random.default_rng(7)
    worst = 0.0
    for cutoff in (0.9, 1.2, 1.5):
        cut2 = np.float32(cutoff * cutoff)
        for extent in (cutoff + 0.4, 2.5, 3.999):
            for with_w in (False, True):
                n = 400_000
                # displacement r - c: a random direction at +-0.03 % of the cutoff (where rounding could matter); the row then
                # anywhere in the range that keeps |r_i| + |c_i| below the extent, component by component
                direction = rng.normal(size=(n, 4))
                if not with_w:
                    direction[:, 3] = 0.0
                direction /= np.linalg.norm(direction, axis=1, keepdims=True)
                dist = cutoff * (1.0 + rng.uniform(-3e-4, 3e-4, n))
                delta = direction * dist[:, None]
                lo, hi = 0.5 * (delta - extent), 0.5 * (delta + extent)
                mid, half = 0.5 * (lo + hi), 0.5 * (hi - lo) * 0.999
                r = f32(mid + rng.uniform(-1.0, 1.0, (n, 4)) * half)
                if not with_w:
                    r[:, 3] = 0.0
                c = f32(r.astype(np.float64) - delta)
                ok = np.all(np.abs(r) + np.abs(c) < extent, axis=1)
                r, c = r[ok], c[ok]
                assert len(r) > 1000
                d2_exact = ((r.astype(np.float64) - c.astype(np.float64)) ** 2).sum(axis=1)  # exact enough: f64 on f32 inputs
                inside = d2_exact < float(cut2)
                accepted = gram_accepts(r, c, cut2, with_w)
                assert np.all(accepted[inside]), (cutoff, extent, with_w, int(np.sum(inside & ~accepted)))
                # and the form is not needlessly loose: nothing beyond cut2 + 2e-4 passes
                assert not np.any(accepted & (d2_exact > float(cut2) + 2e-4))
                # the largest rounding error seen, against the bound of the kernel's comment (3.1e-5 at extent 4)
                rq = (r.astype(np.float64) ** 2).sum(axis=1)
                cc = (c.astype(np.float64) ** 2).sum(axis=1)
                worst = max(worst, float(np.max(np.abs((rq - 2.0 * (r.astype(np.float64) * c.astype(np.float64)).sum(axis=1) + cc) - d2_exact))))
    assert worst < 1e-9  # (the identity itself, in f64: sanity of the test)
