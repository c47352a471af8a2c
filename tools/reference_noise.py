"""
How far is the reference's float32/float64 evaluation from exact arithmetic?  (why the 1e-5
tolerance cannot be met by a differently-rounded kernel.)  Runs the numpy oracle (= the reference's arithmetic) and a
float64-everywhere evaluation of the same formulas on four kinds of 5x5-kernel data and prints the share of pixels whose
gain / corrected value differ by more than 1e-5 relative.  CPU only:  python tools/reference_noise.py
"""
import os
import sys

import numpy as np
from numpy.lib.stride_tricks import sliding_window_view as swv

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle_np as onp  # noqa: E402


def exact_gain_offset(src, ref, k):
    s, r = src.astype(np.float64), ref.astype(np.float64)
    win = lambda a: swv(a, (k, k)).sum(axis=(2, 3))  # noqa: E731
    n, ss, rs, ps, s2 = k * k, win(s), win(r), win(s * r), win(s * s)
    g = (n * ps - ss * rs) / (n * s2 - ss * ss)
    return g, (rs - g * ss) / n


def main():
    rng = np.random.default_rng(0)
    h, w, k = 200, 300, 5
    cases = {'synthetic bench data, U[0.05,1)': None, 'DN 1000 +- 150': (1000, 150, 5), 'DN 1000 +- 20': (1000, 20, 2),
             'DN 5000 +- 25': (5000, 25, 3)}
    for name, cfg in cases.items():
        if cfg is None:
            src, ref = onp.synth_pair(h, w, 1)
        else:
            src = rng.normal(cfg[0], cfg[1], (h, w)).astype(np.float32)
            ref = (1.2 * src.astype(np.float64) + 50 + rng.normal(0, cfg[2], (h, w))).astype(np.float32)
        p, _ = onp.fit_gain_offset(src, None, ref, None, (k, k), False, None)
        r = k // 2
        g_ref, o_ref = p[0][r:-r, r:-r].astype(np.float64), p[1][r:-r, r:-r].astype(np.float64)
        g, o = exact_gain_offset(src, ref, k)
        s_in = src[r:-r, r:-r]
        rel_g = np.abs(g_ref - g) / np.abs(g)
        rel_c = np.abs((g_ref * s_in + o_ref) - (g * s_in + o)) / np.abs(g * s_in + o)
        print(f'{name:32s} gain: median {np.median(rel_g):.1e}, > 1e-5 on {100 * (rel_g > 1e-5).mean():5.1f} % of pixels;'
              f'  corrected: median {np.median(rel_c):.1e}, > 1e-5 on {100 * (rel_c > 1e-5).mean():5.1f} %')


if __name__ == '__main__':
    main()
