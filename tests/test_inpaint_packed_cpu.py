"""
CPU test: the PACKED in-painting search of hk_inpaint.hip (fill_fast) as a numpy model against the oracle's restatement
of GDALFillNodata (oracle_np.fill_nodata; reference call site kernel_model.py:366).

The device search keeps 16-bit keys per quadrant -- (squared distance << 5) | column distance for the first candidate met at
the best distance, the complement of the column distance in the low bits for the last -- over row distances clipped at
FAST_CLIP, declares a quadrant SETTLED after step S when its best squared distance is below (S + 1)^2, and hands every
target with an unsettled quadrant after FAST_LAST to the general search.  The model below restates exactly that (same
constants, same order of the weighted sums); wherever it settles it must reproduce the oracle bit for bit -- ties between
columns at equal distance and GDAL's float comparison of them included -- and on dense source masks it must settle nearly
everywhere.  The constants are read from the kernel source so that the model cannot drift from it.
"""
import os
import re

import numpy as np
import pytest

from oracle import oracle_np as onp

SRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'homonim_amd', 'csrc', 'hk_inpaint.hip')


def _const(name):
    text = open(SRC).read()
    m = re.search(r'constexpr\s+(?:int|unsigned)\s+' + name + r'\s*=\s*(\w+)\s*;', text)
    assert m, name
    val = m.group(1)
    if not val.isdigit():  # defined through a macro with a default
        m2 = re.search(r'#define\s+' + val + r'\s+(\d+)', text)
        assert m2, val
        val = m2.group(1)
    return int(val)


FAST_LAST, FAST_CLIP, MAX_DIST = _const('FAST_LAST'), _const('FAST_CLIP'), _const('FILL_MAX_DIST')
NONE_SQ = 0x7fff


def test_constants_of_the_packed_search():
    assert (FAST_CLIP + FAST_LAST ** 2) * 32 + 31 <= 0xffff          # a key fits its half
    assert FAST_CLIP > (FAST_LAST + 1) ** 2                           # a clipped candidate cannot settle a quadrant
    assert FAST_LAST <= 31 and FAST_LAST % 4 == 0 and MAX_DIST == 100  # the column distance fits five bits; groups of four


def column_tables(src_mask, max_dist=MAX_DIST):
    """ squared row distances to the nearest source at-or-above / strictly below, NONE_SQ beyond max_dist (inpaint_table_kernel) """
    h, w = src_mask.shape
    up = np.full((h, w), NONE_SQ, np.int64)
    dn = np.full((h, w), NONE_SQ, np.int64)
    for x in range(w):
        last = None
        for y in range(h):
            if src_mask[y, x]:
                last = y
            if last is not None and y - last <= max_dist:
                up[y, x] = (y - last) ** 2
        last = None
        for y in range(h - 1, -1, -1):
            if last is not None and last - y <= max_dist + 1:
                dn[y, x] = (last - y) ** 2
            if src_mask[y, x]:
                last = y
    return up, dn


def packed_fill(image, src_mask):
    """ (filled image, settled mask): the packed search of every target; unsettled targets keep their value """
    h, w = image.shape
    up, dn = column_tables(src_mask)
    out = image.copy()
    settled = np.zeros((h, w), bool)
    clip = lambda sq: min(int(sq), FAST_CLIP) << 5  # noqa: E731  (fast_stage_word)
    for y in range(h):
        for x in range(w):
            if src_mask[y, x]:
                continue
            kf = [0xffff] * 4  # quadrants: 0 up-left, 1 down-left (both with the own column), 2 up-right, 3 down-right
            kl = [0xffff] * 4
            ok = False
            for k in range(FAST_LAST + 1):
                cf, cl = (k * k << 5) + k, (k * k << 5) + 31 - k
                for side, xs in ((0, x - k), (2, x + k)):
                    if (side == 2 and k == 0) or xs < 0 or xs >= w:   # outside the raster: no source (GDAL re-checks the edge column)
                        continue
                    for q, sq in ((side, up[y, xs]), (side + 1, dn[y, xs])):
                        kf[q] = min(kf[q], clip(sq) + cf)
                        kl[q] = min(kl[q], clip(sq) + cl)
                        assert clip(sq) + cl <= 0xffff
                if k >= 4 and k % 4 == 0 and max(kf) >> 5 < (k + 1) ** 2:
                    ok = True
                    break
            if not ok:
                continue
            wsum, vsum = np.float64(0), np.float64(0)
            for q in range(4):
                n = kf[q] >> 5
                assert n == kl[q] >> 5 and 0 < n < (FAST_LAST + 1) ** 2
                root = np.sqrt(np.float64(n))
                tie = root * root > n                                  # GDAL's QUAD_CHECK on an equal squared distance
                dx = 31 - (kl[q] & 31) if tie else kf[q] & 31
                dy = int(round(np.sqrt(n - dx * dx)))
                assert dy * dy + dx * dx == n
                sy, sx = (y + dy if q & 1 else y - dy), (x + dx if q >= 2 else x - dx)
                assert src_mask[sy, sx]
                wgt = np.float64(1) / root
                wsum = wsum + wgt
                vsum = vsum + np.float64(image[sy, sx]) * wgt
            out[y, x] = np.float32(vsum / wsum)
            settled[y, x] = True
    return out, settled


@pytest.mark.parametrize('h, w, density, seed', [(40, 70, 0.65, 1), (40, 70, 0.3, 2), (48, 64, 0.06, 3), (30, 90, 0.9, 4),
                                                 (64, 64, 0.02, 5)])
def test_packed_search_equals_the_restatement_where_it_settles(h, w, density, seed):
    rng = np.random.default_rng(seed)
    img = rng.normal(0, 1, (h, w)).astype(np.float32)
    src = rng.uniform(size=(h, w)) < density
    exp = onp.fill_nodata(img, src)
    got, settled = packed_fill(img, src)
    assert (got[settled] == exp[settled]).all(), np.argwhere(settled & (got != exp))[:5]
    assert (got[src] == img[src]).all()
    inner = np.zeros((h, w), bool)
    inner[8:-8, 8:-8] = True
    if density >= 0.3:   # dense sources: everything away from the raster's edges settles
        assert settled[inner & ~src].all()
    assert settled.sum() > 0


def test_packed_search_ties_between_columns():
    """ sources placed so that several columns offer the same squared distance (5-12-13 and 3-4-5 triangles, diagonals): the tie
    rule of GDAL's float comparison decides which source's value is taken """
    h, w = 41, 61
    img = np.arange(h * w, dtype=np.float32).reshape(h, w)
    src = np.zeros((h, w), bool)
    cy, cx = 20, 30
    for dy, dx in ((3, 4), (4, 3), (5, 0), (0, 5), (5, 12), (12, 5), (13, 0), (1, 7), (7, 1), (5, 5), (2, 2), (2, 11), (10, 5), (11, 2)):
        for sy, sx in ((cy - dy, cx - dx), (cy - dy, cx + dx), (cy + dy, cx - dx), (cy + dy, cx + dx)):
            src[sy, sx] = True
    src[cy, cx] = False
    exp = onp.fill_nodata(img, src)
    got, settled = packed_fill(img, src)
    assert settled[cy, cx]
    assert (got[settled] == exp[settled]).all()
    assert settled.sum() > 300
