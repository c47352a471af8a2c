""" PROOFS.md appendix C on the CPU: float32(t / N) == float32(float64(t) * RN64(1 / N)) for window counts N < 2^16 -- the form the fused
kernel computes the offset division of kernel_model.py:351 in (hk_fit_kernel.h HK_INV_N, FitArgs::inv_n_full). """
import numpy as np


def test_reciprocal_multiplication_equals_the_float32_division():
    rng = np.random.default_rng(7)
    n = np.arange(1, 65536, dtype=np.int64)
    inv = 1.0 / n.astype(np.float64)                                   # RN64(1 / N)
    for _ in range(24):
        # mantissas over the whole range, exponents that keep the quotient a normal float32, both signs
        m = rng.integers(1 << 23, 1 << 24, size=n.size).astype(np.float64)
        e = rng.integers(-60, 60, size=n.size)
        t = (np.ldexp(m, e) * rng.choice([-1.0, 1.0], size=n.size)).astype(np.float32)
        want = t / n.astype(np.float32)                                # IEEE float32 division
        got = (t.astype(np.float64) * inv).astype(np.float32)
        assert np.array_equal(want.view(np.uint32), got.view(np.uint32))
    # the adversarial mantissas of step 2: quotients as close to a rounding boundary as an integer N allows (t = N * (m + 1/2) rounded)
    for N in (3, 25, 255, 289, 961, 1023, 4097, 49215, 65535):
        mm = rng.integers(1 << 23, 1 << 24, size=200000).astype(np.float64) + 0.5
        t = np.float32(1) * (mm * N).astype(np.float32)                # near-boundary numerators, rounded to float32 either way
        for tt in (t, np.nextafter(t, np.float32(np.inf)), np.nextafter(t, np.float32(-np.inf))):
            want = tt / np.float32(N)
            got = (tt.astype(np.float64) * (1.0 / N)).astype(np.float32)
            assert np.array_equal(want.view(np.uint32), got.view(np.uint32)), N
