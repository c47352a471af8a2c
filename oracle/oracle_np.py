"""
ORACLE -- TEST INFRASTRUCTURE ONLY.  Not part of the product path.

numpy restatement of the homonim sliding-window kernel-model fit/apply hot path
(reference: /root/reference/homonim/kernel_model.py @ v0.4.3).  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this module; the
product (``homonim_amd``) never does.

What is restated, and from where
--------------------------------
* ``nan_equals``            <- homonim/utils.py:54-56
* ``mask_of``               <- homonim/raster_array.py:298-308 (``RasterArray.mask``)
* ``box_sum``               <- the call sites kernel_model.py:167-175,184,257-258,332-341 of
                               ``cv.boxFilter`` / ``cv.sqrBoxFilter(normalize=False, BORDER_CONSTANT)``.
                               OpenCV (opencv-python-headless>=4.5, un-pinned, pyproject.toml:8) is NOT in
                               /root/reference and not installed here, so its published algorithm is
                               restated: zero-padded kh x kw window sum anchored at the centre, accumulated
                               in float64 (OpenCV picks sumType=CV_64F for CV_32F/CV_64F input).  boxFilter
                               casts the result to the input depth; sqrBoxFilter squares in float64 and RETURNS
                               float64 (its ddepth=-1 rule), see ``box_sum``.
* ``r2_array``              <- kernel_model.py:142-214
* ``fit_block_norm``        <- kernel_model.py:216-229
* ``fit_gain``              <- kernel_model.py:231-274
* ``fit_gain_blk_offset``   <- kernel_model.py:276-303   (NumPy>=2 flavour: normalised source is float64)
* ``fit_gain_offset``       <- kernel_model.py:305-373   (in-paint branch: bookkeeping only, see below)
* ``apply``                 <- kernel_model.py:442-463
* ``compare_sums``, ``compare_stats`` <- compare.py:243-255 (get_block_sums), :142-186 (_get_image_stats)

Pinning
-------
The Python part of this restatement is pinned bit-for-bit against golden vectors produced by executing the
reference's own ``kernel_model.py`` in the build container (``oracle/gen_golden.py`` -> ``tests/golden/*.npz``).
The OpenCV boundary itself (float64 accumulation, float64 sqrBoxFilter output) is pinned by the reference's own
PARAM GeoTIFF (``tests/golden/ref_param_tif.npz``, written by the real homonim+OpenCV+GDAL stack), which this
restatement reproduces bit-for-bit, and by the reference tests' known answers -- see DESIGN.md.

All functions work on plain ndarrays + nodata scalars (no RasterArray), never mutate their inputs (they copy), and
keep the reference's operation order so results are bitwise reproducible.
"""
from typing import Optional, Tuple

import math

import numpy as np

F32 = np.float32
NAN32 = np.float32(np.nan)


def nan_equals(a, b):
    """ utils.py:54-56 """
    return (a == b) | (np.isnan(a) & np.isnan(b))


def mask_of(array: np.ndarray, nodata) -> np.ndarray:
    """ raster_array.py:298-308 -- valid-pixel mask of a 2-D array. """
    if nodata is None:
        return np.full(array.shape[-2:], True)
    return ~nan_equals(array, nodata)


def box_sum(x: np.ndarray, kernel_shape: Tuple[int, int], square: bool = False) -> np.ndarray:
    """
    cv.boxFilter / cv.sqrBoxFilter (normalize=False, borderType=BORDER_CONSTANT, ddepth=-1) restated.
    ``kernel_shape`` is (kh, kw) (the reference passes ksize=(kw, kh)=kernel_shape[::-1]).
    Accumulates in float64.  ``square=False`` (boxFilter) returns ``x.dtype``; ``square=True`` (sqrBoxFilter)
    returns float64: OpenCV resolves ``ddepth=-1`` there as ``sdepth < CV_32F ? CV_32F : CV_64F``, so a float32
    input yields a float64 sum-of-squares and everything downstream of it (m_den, the gain division, all of R2)
    runs in float64 under numpy promotion.  Only this reproduces the reference's PARAM GeoTIFF bit-for-bit
    (tests/test_oracle_golden.py::test_reference_param_tif).
    """
    kh, kw = int(kernel_shape[0]), int(kernel_shape[1])
    rh, rw = kh // 2, kw // 2
    h, w = x.shape
    xd = x.astype(np.float64)
    if square:
        xd = xd * xd
    pad = np.zeros((h + 2 * rh, w + 2 * rw), dtype=np.float64)
    pad[rh:rh + h, rw:rw + w] = xd
    # separable direct sums (order is immaterial while the float64 partial sums are exact -- DESIGN.md)
    rows = np.zeros((h + 2 * rh, w), dtype=np.float64)
    for dx in range(kw):
        rows += pad[:, dx:dx + w]
    out = np.zeros((h, w), dtype=np.float64)
    for dy in range(kh):
        out += rows[dy:dy + h, :]
    return out if square else out.astype(x.dtype)


def r2_array(
    ref_array, src_array, param_array, mask=None, mask_sum=None, ref_sum=None, src_sum=None, ref2_sum=None,
    src2_sum=None, src_ref_sum=None, dest_array=None, kernel_shape=(5, 5)
):
    """ kernel_model.py:142-214.  NOTE: like the reference, zero-fills ref/src in place when ``mask`` is None. """
    if mask is None:
        mask = ~nan_equals(src_array, np.nan) & ~nan_equals(ref_array, np.nan)  # :160-163
        ref_array[~mask] = 0
        src_array[~mask] = 0
    if mask_sum is None:
        mask_sum = box_sum(mask.astype(F32), kernel_shape)  # :167
    if ref_sum is None:
        ref_sum = box_sum(ref_array, kernel_shape)  # :169
    if ref2_sum is None:
        ref2_sum = box_sum(ref_array, kernel_shape, square=True)  # :171
    if src2_sum is None:
        src2_sum = box_sum(src_array, kernel_shape, square=True)  # :173
    if src_ref_sum is None:
        src_ref_sum = box_sum(src_array * ref_array, kernel_shape)  # :175

    ss_tot_array = (mask_sum * ref2_sum) - (ref_sum ** 2)  # :179

    if param_array.shape[0] > 1:
        if src_sum is None:
            src_sum = box_sum(src_array, kernel_shape)  # :184
        ss_res_array = (
            ((param_array[0] ** 2) * src2_sum) +
            (2 * np.prod(param_array[:2], axis=0) * src_sum) -
            (2 * param_array[0] * src_ref_sum) -
            (2 * param_array[1] * ref_sum) +
            ref2_sum + (mask_sum * (param_array[1] ** 2))
        )  # :189-195
    else:
        ss_res_array = (((param_array[0] ** 2) * src2_sum) - (2 * param_array[0] * src_ref_sum) + ref2_sum)  # :201

    ss_res_array *= mask_sum  # :203

    if dest_array is None:
        dest_array = np.full(src_array.shape, fill_value=np.nan, dtype=F32)  # :207-209

    with np.errstate(all='ignore'):
        np.divide(ss_res_array, ss_tot_array, out=dest_array, where=mask)  # :212
        np.subtract(1, dest_array, out=dest_array, where=mask)  # :213
    return dest_array


def fit_block_norm(src, src_nodata, ref, ref_nodata) -> np.ndarray:
    """ kernel_model.py:216-229 """
    norm_model = np.zeros(2)
    mask = mask_of(ref, ref_nodata) & mask_of(src, src_nodata)
    if not np.any(mask):
        return norm_model
    with np.errstate(all='ignore'):
        norm_model[0] = np.std(ref[mask]) / np.std(src[mask])
        norm_model[1] = np.percentile(ref[mask], 1) - np.percentile(src[mask], 1) * norm_model[0]
    return norm_model


# ---- the same statistics for a block whose rows are spread over several ranks (homonim_amd/split_norm.py, hk_norm.hip
# launch_block_norm_split): every ingredient is a sum over pixels, so each rank contributes arrays that are all-reduced (SUM)
# between the phases.  This is the numpy statement of what each phase contributes and of what is done with the reduced
# arrays -- the order statistics come out of an exact 3-level radix select on the order-preserving uint32 image of float32.
SPLIT_BITS = (11, 11, 10)


def _f2key(v):
    u = np.ascontiguousarray(v, F32).view(np.uint32)
    return np.where(u & np.uint32(0x80000000), ~u, u | np.uint32(0x80000000)).astype(np.uint32)


def _key2f(k):
    k = np.uint32(k)
    u = (k & np.uint32(0x7fffffff)) if (k & np.uint32(0x80000000)) else np.uint32(~k)
    return np.array([u], np.uint32).view(F32)[0]


def split_norm_slab_values(src, src_nodata, ref, ref_nodata):
    """ the slab's jointly valid (src, ref) values """
    mask = mask_of(ref, ref_nodata) & mask_of(src, src_nodata)
    return src[mask].astype(F32), ref[mask].astype(F32)


def split_norm_moments(vals, shift):
    """ phase 1 contribution: [n, sum(s - shift_s), sum((s - shift_s)^2), sum(r - shift_r), sum((r - shift_r)^2)] """
    s, r = (v.astype(np.float64) for v in vals)
    ds, dr = s - shift[0], r - shift[1]
    return np.array([s.size, ds.sum(), (ds * ds).sum(), dr.sum(), (dr * dr).sum()])


def split_norm_hist(vals, level, prefixes):
    """ phases 2-4 contribution: histograms [raster][rank] of the level's key digit over the values whose higher digits
    equal the rank's prefix (level 0: every value; both ranks of a raster share one histogram) """
    out = np.zeros((2, 2, 1 << SPLIT_BITS[level]))
    for q in range(2):
        keys = _f2key(vals[q])
        for k in range(2):
            if level == 0:
                if k == 0:
                    out[q, 0] = np.bincount(keys >> np.uint32(21), minlength=2048)
            elif level == 1:
                sel = (keys >> np.uint32(21)) == prefixes[q][k]
                out[q, k] = np.bincount((keys[sel] >> np.uint32(10)) & np.uint32(2047), minlength=2048)
            else:
                sel = (keys >> np.uint32(10)) == prefixes[q][k]
                out[q, k] = np.bincount(keys[sel] & np.uint32(1023), minlength=1024)
    return out


def split_norm_select(hist, level, prefixes, ranks):
    """ with the REDUCED histograms: the bin that holds each rank, the new prefixes and the ranks inside those bins """
    new_p, new_r = [[0, 0], [0, 0]], [[0, 0], [0, 0]]
    for q in range(2):
        for k in range(2):
            h = hist[q, 0 if level == 0 else k]
            cum = np.cumsum(h)
            b = int(np.searchsorted(cum, ranks[q][k], side='right'))
            b = min(b, h.size - 1)
            new_p[q][k] = b if level == 0 else (prefixes[q][k] << SPLIT_BITS[level]) | b
            new_r[q][k] = ranks[q][k] - (int(cum[b - 1]) if b else 0)
    return new_p, new_r


def split_norm_finish(moments, shift, prefixes):
    """ norm from the reduced moments and the two selected order statistics per raster (numpy's _lerp for the percentile) """
    n = int(moments[0])
    if n == 0:
        return np.zeros(2)
    var = []
    for q in range(2):
        d = moments[1 + 2 * q] / n
        var.append(max(moments[2 + 2 * q] / n - d * d, 0.0))
    vi = 0.01 * (n - 1)
    t = vi - math.floor(vi)
    pct = []
    for q in range(2):
        lo, hi = float(_key2f(prefixes[q][0])), float(_key2f(prefixes[q][1]))
        d = hi - lo
        pct.append(hi - d * (1 - t) if t >= 0.5 else lo + d * t)
    n0 = math.sqrt(var[1]) / math.sqrt(var[0])
    return np.array([n0, pct[1] - pct[0] * n0])


def split_norm_ranks(n):
    vi = 0.01 * (n - 1)
    k0 = int(math.floor(vi))
    return [[k0, min(k0 + 1, n - 1)], [k0, min(k0 + 1, n - 1)]]


def _fit_gain_arrays(src_array, src_mask, ref_array, ref_mask, kernel_shape, find_r2):
    """ kernel_model.py:231-274 on already-copied arrays (src may be float64 for gain-blk-offset). """
    mask = ref_mask & src_mask  # :245
    ref_array[~mask] = 0  # :246
    src_array[~mask] = 0  # :247
    src_sum = box_sum(src_array, kernel_shape)  # :257
    ref_sum = box_sum(ref_array, kernel_shape)  # :258
    param = np.full((3 if find_r2 else 2, *src_array.shape), np.nan, dtype=F32)  # :261
    param[1, mask] = 0  # :262
    with np.errstate(all='ignore'):
        np.divide(ref_sum, src_sum, out=param[0], where=mask)  # :265
    if find_r2:
        r2_array(
            ref_array, src_array, param[:1], mask=mask, ref_sum=ref_sum, src_sum=src_sum, dest_array=param[2],
            kernel_shape=kernel_shape
        )  # :269-272
    return param


def fit_gain(src, src_nodata, ref, ref_nodata, kernel_shape=(5, 5), find_r2=False) -> np.ndarray:
    """ Model.gain: kernel_model.py:231-274.  Returns params (2|3, H, W) float32. """
    src_array = np.array(src, dtype=F32, copy=True)
    ref_array = np.array(ref, dtype=F32, copy=True)
    return _fit_gain_arrays(
        src_array, mask_of(src_array, src_nodata), ref_array, mask_of(ref_array, ref_nodata), kernel_shape, find_r2
    )


def fit_gain_blk_offset(
    src, src_nodata, ref, ref_nodata, kernel_shape=(5, 5), find_r2=False, norm_model: Optional[np.ndarray] = None
) -> Tuple[np.ndarray, np.ndarray]:
    """
    Model.gain_blk_offset: kernel_model.py:276-303, NumPy>=2 flavour (``src * np.float64`` promotes to float64).
    ``norm_model`` may be injected (float64[2]) to isolate the window part from the block statistics.
    Returns (params, norm_model).
    """
    src_array = np.array(src, dtype=F32, copy=True)
    ref_array = np.array(ref, dtype=F32, copy=True)
    if norm_model is None:
        norm_model = fit_block_norm(src_array, src_nodata, ref_array, ref_nodata)  # :289
    norm_model = np.asarray(norm_model, dtype=np.float64)
    # src_ra.nodata = nan (:292; raster_array.py:334-351): masked pixels become nan, nodata None -> nan w/o change
    if src_nodata is not None and not nan_equals(np.nan, src_nodata):
        src_array[~mask_of(src_array, src_nodata)] = np.nan
    with np.errstate(all='ignore'):
        src_norm = (src_array * norm_model[0]) + norm_model[1]  # :295 -- float64 under NumPy>=2
    assert src_norm.dtype == np.float64
    param = _fit_gain_arrays(
        src_norm, mask_of(src_norm, np.nan), ref_array, mask_of(ref_array, ref_nodata), kernel_shape, find_r2
    )  # :298
    with np.errstate(all='ignore'):
        param[1] = param[0] * norm_model[1]  # :301
        param[0] *= norm_model[0]  # :302
    return param, norm_model


def fill_nodata(image: np.ndarray, mask: np.ndarray, max_search_distance: float = 100.0) -> np.ndarray:
    """
    rasterio.fill.fillnodata(image, mask, max_search_distance=100, smoothing_iterations=0) == GDALFillNodata, which the
    reference calls at kernel_model.py:366.  GDAL is NOT in /root/reference (rasterio>=1.1, un-pinned, pyproject.toml:7)
    and not installed here: its published algorithm (gdal/alg/rasterfill.cpp) is restated, PARITY WITH GDAL UNPINNED.
    Pixels with mask == 0 are interpolated from pixels with mask != 0: per target, the nearest source in each of four
    quadrants (found through per-column nearest-above / nearest-below tables while stepping outwards one column at a
    time), inverse-distance weighted.  Plain loops -- small inputs only.
    """
    h, w = image.shape
    src = mask.astype(bool)
    md = int(np.floor(max_search_distance))
    none = np.iinfo(np.int64).max
    top_y = np.full((h, w), none, np.int64)
    top_v = np.zeros((h, w), np.float32)
    bot_y = np.full((h, w), none, np.int64)
    bot_v = np.zeros((h, w), np.float32)
    for x in range(w):
        last_y, last_v = none, np.float32(0)
        for y in range(h):                      # top-down: nearest source at or above
            if src[y, x]:
                last_y, last_v = y, image[y, x]
            elif last_y != none and y > md + last_y:
                last_y = none
            top_y[y, x], top_v[y, x] = last_y, last_v
        last_y, last_v = none, np.float32(0)
        for y in range(h - 1, -1, -1):          # bottom-up: nearest source strictly below
            bot_y[y, x], bot_v[y, x] = last_y, last_v
            if src[y, x]:
                last_y, last_v = y, image[y, x]
            elif last_y != none and last_y - y > md:
                last_y = none
    out = image.copy()
    for y in range(h):
        for x in range(w):
            if src[y, x]:
                continue
            qd = [max_search_distance + 1.0] * 4
            qv = [0.0] * 4

            def check(q, tx, ty, tv):
                if ty == none:
                    return
                d2 = float(tx - x) ** 2 + float(ty - y) ** 2
                if d2 < qd[q] * qd[q]:
                    qd[q] = float(np.sqrt(d2))
                    qv[q] = float(tv)

            this_max = md
            step = 0
            while step <= this_max:
                lx, rx = max(0, x - step), min(w - 1, x + step)
                check(0, lx, top_y[y, lx], top_v[y, lx])
                check(1, lx, bot_y[y, lx], bot_v[y, lx])
                if step > 0:
                    check(2, rx, top_y[y, rx], top_v[y, rx])
                    check(3, rx, bot_y[y, rx], bot_v[y, rx])
                    if step & 3 == 0:
                        this_max = int(np.floor(max(qd)))
                step += 1
            wsum = vsum = 0.0
            has = False
            for q in range(4):
                if qd[q] <= max_search_distance:
                    wgt = 1.0 / qd[q]
                    has = wgt != 0
                    wsum += wgt
                    vsum += qv[q] * wgt
            if has:
                out[y, x] = np.float32(vsum / wsum)
    return out


def fit_gain_offset(
    src, src_nodata, ref, ref_nodata, kernel_shape=(5, 5), find_r2=False, r2_inpaint_thresh: Optional[float] = 0.25
) -> Tuple[np.ndarray, int]:
    """
    Model.gain_offset: kernel_model.py:305-373.
    Returns (params, n_fail) where n_fail is the number of valid pixels failing the r2 mask test (:363).  When
    n_fail == 0 GDAL's ``fillnodata`` (:366) is the identity on valid pixels and the result is pinned by the reference
    goldens; when n_fail > 0 the in-painting runs through ``fill_nodata`` above, a restatement of GDAL's published
    algorithm whose parity with GDAL itself is unpinned (no GDAL here).
    """
    src_array = np.array(src, dtype=F32, copy=True)
    ref_array = np.array(ref, dtype=F32, copy=True)
    mask = mask_of(ref_array, ref_nodata) & mask_of(src_array, src_nodata)  # :319
    ref_array[~mask] = 0
    src_array[~mask] = 0
    _find_r2 = find_r2 or (r2_inpaint_thresh is not None)  # :325

    src_sum = box_sum(src_array, kernel_shape)  # :332
    ref_sum = box_sum(ref_array, kernel_shape)  # :333
    src_ref_sum = box_sum(src_array * ref_array, kernel_shape)  # :334
    mask_sum = box_sum(mask.astype(F32), kernel_shape)  # :335-337
    with np.errstate(all='ignore'):
        m_num_array = (mask_sum * src_ref_sum) - (src_sum * ref_sum)  # :338
        src2_sum = box_sum(src_array, kernel_shape, square=True)  # :341
        m_den_array = (mask_sum * src2_sum) - (src_sum ** 2)  # :342

        param = np.full((3 if _find_r2 else 2, *src_array.shape), np.nan, dtype=F32)  # :345
        np.divide(m_num_array, m_den_array, out=param[0], where=mask)  # :348
        np.divide(ref_sum - (param[0] * src_sum), mask_sum, out=param[1], where=mask)  # :351

        if _find_r2:
            r2_array(
                ref_array, src_array, param[:2], mask=mask, mask_sum=mask_sum, ref_sum=ref_sum, src_sum=src_sum,
                src2_sum=src2_sum, src_ref_sum=src_ref_sum, dest_array=param[2], kernel_shape=kernel_shape
            )  # :355-359

        n_fail = 0
        if r2_inpaint_thresh is not None:
            r2_mask = (param[2] > r2_inpaint_thresh) & (param[0] > 0) & mask  # :363
            n_fail = int(np.count_nonzero(~r2_mask & mask))
            if n_fail == 0:
                # fillnodata(offset, r2_mask) leaves r2_mask pixels untouched and fills the rest (all of which are
                # ~mask here); ``param_ra.mask = mask`` (:367) then resets every band at ~mask to nan.
                param[:, ~mask] = np.nan
                # :370-371 is a no-op (its where-mask ``~r2_mask & mask`` is empty)
            else:
                param[1] = fill_nodata(param[1], r2_mask)  # :366
                param[:, ~mask] = np.nan  # :367
                redo = ~r2_mask & mask  # :370
                np.divide(ref_sum - mask_sum * param[1], src_sum, out=param[0], where=redo)  # :371
    return param, n_fail


def full_coverage_mask(in_mask: np.ndarray, params: np.ndarray, kernel_shape) -> np.ndarray:
    """
    kernel_model.py:375-409 on a shared grid (the two re-projections are identities there): in_mask & param mask
    (any of the gain / offset bands not nodata), eroded by a (kh+2) x (kw+2) rectangle with a zero border.
    """
    mask = in_mask.astype(bool) & np.any(~np.isnan(params[:2]), axis=0)  # :399-401
    kh, kw = int(kernel_shape[0]) + 2, int(kernel_shape[1]) + 2          # :407
    cnt = box_sum(mask.astype(F32), (kh, kw))                             # erosion by a full rectangle == full count
    return cnt == kh * kw


def apply(src, param) -> np.ndarray:
    """ kernel_model.py:461 -- two float32 roundings, no FMA. """
    with np.errstate(all='ignore'):
        return (param[0] * np.asarray(src, dtype=F32)) + param[1]


def fit(model: str, src, src_nodata, ref, ref_nodata, kernel_shape=(5, 5), find_r2=False, r2_inpaint_thresh=0.25,
        norm_model=None):
    """ kernel_model.py:411-440 dispatch.  Returns (params, aux) with aux = norm_model | n_fail | None. """
    if model == 'gain':
        return fit_gain(src, src_nodata, ref, ref_nodata, kernel_shape, find_r2), None
    elif model == 'gain-blk-offset':
        return fit_gain_blk_offset(src, src_nodata, ref, ref_nodata, kernel_shape, find_r2, norm_model)
    elif model == 'gain-offset':
        return fit_gain_offset(src, src_nodata, ref, ref_nodata, kernel_shape, find_r2, r2_inpaint_thresh)
    raise ValueError(model)


# ----------------------------------------------------------------------------------------------------------------------
# Re-sampling around the path: RasterArray.reproject (raster_array.py:526-578) -> rasterio.warp.reproject -> GDAL warp.
# GDAL is NOT in /root/reference (rasterio>=1.1, un-pinned, pyproject.toml:7) and not installed here: the published warp
# kernels (gdal/alg/gdalwarpkernel.cpp) are restated for the only geometry the block pipeline produces -- same CRS,
# north-up axis-aligned grids -- PARITY WITH GDAL UNPINNED.  Anchors: the reference tests' known answers for the aligned
# 2:1 pair (tests/test_kernel_model.py:32-117) and resampling invariants (tests/test_oracle_golden.py).
#
#   mapping  : (kx, ox, ky, oy) with src_col = kx * dst_col + ox, src_row = ky * dst_row + oy on continuous pixel
#              coordinates whose integers are pixel EDGES (pixel i covers [i, i+1)).
#   nearest  : GWKNearest -- the source pixel containing the destination pixel centre (floor(x + 1e-10)).
#   average  : GWKAverageOrMode -- weighted mean of the valid source pixels overlapping the destination pixel's footprint,
#              weight = fractional overlap per axis; nodata when no valid pixel.
#   mode / med / q1 / q3 : the same function's rank-order branches over the same footprint, unweighted (see the code).
#   bilinear / cubic_spline : GWKResample -- the source pixel under the destination centre must be valid; separable
#              2 / 4-tap kernel (cubic B-spline) around it, taps outside the raster or invalid are skipped and the sum is
#              renormalised by the accumulated weight (nodata below 1e-6).  Up-sampling only (kernel not stretched).
RESAMPLING_CODES = {'nearest': 0, 'bilinear': 1, 'cubic': 2, 'cubic_spline': 3, 'lanczos': 4, 'average': 5, 'mode': 6, 'max': 8,
                    'min': 9, 'med': 10, 'q1': 11, 'q3': 12, 'sum': 13, 'rms': 14}


COMPARE_KEYS = ('src_sum', 'ref_sum', 'src2_sum', 'ref2_sum', 'src_ref_sum', 'res2_sum', 'mask_sum')


def compare_sums(src, src_nodata, ref, ref_nodata) -> dict:
    """ compare.py:243-255 for one block of two rasters on one grid: the seven masked sums, with the reference's dtypes
    (float32 arrays -> numpy float32 pairwise sums; the count is an integer). """
    src_array = np.array(src, dtype=np.float32, copy=True)
    ref_array = np.array(ref, dtype=np.float32, copy=True)
    mask = mask_of(ref_array, ref_nodata) & mask_of(src_array, src_nodata)
    src_array[~mask] = 0
    ref_array[~mask] = 0
    return dict(
        src_sum=src_array.sum(), ref_sum=ref_array.sum(), src2_sum=(src_array ** 2).sum(),
        ref2_sum=(ref_array ** 2).sum(), src_ref_sum=(src_array * ref_array).sum(),
        res2_sum=((ref_array - src_array) ** 2).sum(), mask_sum=mask.sum()
    )


def compare_band_stats(src_sum=0, ref_sum=0, src2_sum=0, ref2_sum=0, src_ref_sum=0, res2_sum=0, mask_sum=0) -> dict:
    """ compare.py:145-163: Pearson r2, RMSE, rRMSE and N of one band from its accumulated sums. """
    src_mean = src_sum / mask_sum
    ref_mean = ref_sum / mask_sum
    pcc_num = src_ref_sum - (mask_sum * src_mean * ref_mean)
    pcc_den = np.sqrt(src2_sum - (mask_sum * (src_mean ** 2))) * np.sqrt(ref2_sum - (mask_sum * (ref_mean ** 2)))
    pcc = pcc_num / pcc_den
    rmse = np.sqrt(res2_sum / mask_sum)
    rrmse = rmse / ref_mean
    return dict(r2=pcc ** 2, rmse=rmse, rrmse=rrmse, n=int(mask_sum))


def compare_stats(image_sums, band_names=None) -> dict:
    """ compare.py:142-186: per-band statistics + their 'Mean' (integers stay integers). """
    image_stats, sum_over_bands = {}, {}
    for band_i, band_sum_dict in enumerate(image_sums):
        band_stats = compare_band_stats(**band_sum_dict)
        image_stats[band_names[band_i] if band_names else f'Ref. band {band_i + 1}'] = band_stats
        sum_over_bands = {k: sum_over_bands.get(k, 0) + v for k, v in band_stats.items()}
    image_stats['Mean'] = {k: int(v / len(image_sums)) if isinstance(v, int) else (v / len(image_sums))
                           for k, v in sum_over_bands.items()}
    return image_stats


def grid_mapping(src_transform, dst_transform):
    """ (kx, ox, ky, oy) between two axis-aligned affine transforms (a, b, c, d, e, f), b = d = 0. """
    sa, sb, sc, sd, se, sf = [float(v) for v in src_transform[:6]]
    da, db, dc, dd, de, df = [float(v) for v in dst_transform[:6]]
    if sb or sd or db or dd:
        raise NotImplementedError('rotated / sheared grids')
    return da / sa, (dc - sc) / sa, de / se, (df - sf) / se


def _bspline_weights(delta):
    """ cubic B-spline weights of taps -1, 0, 1, 2 at fractional offset delta (GWKBSpline). """
    d = np.float64(delta)
    a, b, c = 1.0 - d, 2.0 - d, 3.0 - d  # cubes as explicit products so that every implementation rounds alike
    w0 = a * a * a / 6.0
    w1 = (b * b * b - 4.0 * (a * a * a)) / 6.0
    w2 = (c * c * c - 4.0 * (b * b * b) + 6.0 * (a * a * a)) / 6.0
    w3 = d * d * d / 6.0
    return (w0, w1, w2, w3)


def _conv_weight(kind: str, x: float) -> float:
    """ GDAL's GWKBilinear / GWKCubic (a = -0.5) / GWKBSpline / GWKLanczosSinc (radius 3) at the scaled distance x. """
    ax = abs(x)
    if kind == 'bilinear':
        return 1.0 - ax if ax <= 1.0 else 0.0
    if kind == 'cubic':
        x2 = ax * ax
        if ax <= 1.0:
            return x2 * (1.5 * ax - 2.5) + 1.0
        if ax <= 2.0:
            return x2 * (-0.5 * ax + 2.5) - 4.0 * ax + 2.0
        return 0.0
    if kind == 'cubic_spline':
        if ax > 2.0:
            return 0.0
        xp2, xp1, xm1 = x + 2.0, x + 1.0, x - 1.0
        a = xp2 * xp2 * xp2 if xp2 > 0.0 else 0.0
        b = xp1 * xp1 * xp1 if xp1 > 0.0 else 0.0
        c = x * x * x if x > 0.0 else 0.0
        d = xm1 * xm1 * xm1 if xm1 > 0.0 else 0.0
        return (a - 4.0 * b + 6.0 * c - 4.0 * d) / 6.0
    if kind == 'lanczos':
        if ax >= 3.0:
            return 0.0
        if x == 0.0:
            return 1.0
        px = math.pi * x
        px3 = px / 3.0
        return math.sin(px) * math.sin(px3) / (px * px3)
    raise NotImplementedError(kind)


def _reproject_conv(srcd, valid, mapping, dst_shape, kind):
    """ GWKResample for any scale: taps i in [1 - R', R'] per axis, R' = ceil(R / scale) on a down-sampled axis (scale =
    min(1, 1 / k)), weight f((i - delta) * scale), renormalised by the accumulated weight. """
    kx, ox, ky, oy = mapping
    sh, sw = srcd.shape
    dh, dw = dst_shape
    out = np.zeros((dh, dw), np.float64)
    got = np.zeros((dh, dw), bool)
    R = dict(bilinear=1, cubic=2, cubic_spline=2, lanczos=3)[kind]
    xs = 1.0 / kx if kx > 1.0 else 1.0
    ys = 1.0 / ky if ky > 1.0 else 1.0
    rx = int(math.ceil(R / xs)) if xs < 1.0 else R
    ry = int(math.ceil(R / ys)) if ys < 1.0 else R
    for i in range(dh):
        sy = ky * (i + 0.5) + oy
        cy = int(math.floor(sy + 1e-10))
        iy = int(math.floor(sy - 0.5))
        dy = sy - 0.5 - iy
        for j in range(dw):
            sx = kx * (j + 0.5) + ox
            cx = int(math.floor(sx + 1e-10))
            if cx < 0 or cx >= sw or cy < 0 or cy >= sh or not valid[cy, cx]:
                continue
            ix = int(math.floor(sx - 0.5))
            dx = sx - 0.5 - ix
            acc = wacc = 0.0
            for tj in range(1 - ry, ry + 1):
                a = iy + tj
                if a < 0 or a >= sh:
                    continue
                wy = _conv_weight(kind, (tj - dy) * ys)
                if wy == 0.0:
                    continue
                for ti in range(1 - rx, rx + 1):
                    b = ix + ti
                    if b < 0 or b >= sw or not valid[a, b]:
                        continue
                    wgt = _conv_weight(kind, (ti - dx) * xs) * wy
                    acc += srcd[a, b] * wgt
                    wacc += wgt
            if not abs(wacc) < 1e-6:
                out[i, j] = acc / wacc
                got[i, j] = True
    return out, got


def reproject(src: np.ndarray, src_nodata, mapping, dst_shape, dst_nodata=np.nan, resampling='average',
              dtype=np.float32) -> np.ndarray:
    """ One band (2-D) through the restated GDAL warp kernels; see the block comment above. """
    kx, ox, ky, oy = mapping
    if kx <= 0 or ky <= 0:
        raise NotImplementedError('flipped grids')
    sh, sw = src.shape
    dh, dw = dst_shape
    valid = mask_of(src, src_nodata)
    srcd = src.astype(np.float64)
    fill = 0 if dst_nodata is None else dst_nodata
    out = np.full((dh, dw), fill, dtype=np.float64)
    got = np.zeros((dh, dw), bool)

    if resampling == 'nearest':
        cx = np.floor(kx * (np.arange(dw) + 0.5) + ox + 1e-10).astype(np.int64)
        cy = np.floor(ky * (np.arange(dh) + 0.5) + oy + 1e-10).astype(np.int64)
        inx, iny = (cx >= 0) & (cx < sw), (cy >= 0) & (cy < sh)
        yy, xx = np.meshgrid(np.clip(cy, 0, sh - 1), np.clip(cx, 0, sw - 1), indexing='ij')
        ok = iny[:, None] & inx[None, :] & valid[yy, xx]
        out[ok] = srcd[yy, xx][ok]
        got = ok
    elif resampling in ('cubic', 'lanczos') or (resampling in ('bilinear', 'cubic_spline') and (kx > 1 + 1e-9 or ky > 1 + 1e-9)):
        out, got = _reproject_conv(srcd, valid, mapping, dst_shape, resampling)
    elif resampling in ('average', 'max', 'min', 'sum', 'rms', 'mode', 'med', 'q1', 'q3'):
        for i in range(dh):
            y0, y1 = max(ky * i + oy, 0.0), min(ky * (i + 1) + oy, float(sh))
            iy0, iy1 = int(np.floor(y0 + 1e-10)), int(np.ceil(y1 - 1e-10))
            if iy0 == iy1 and iy1 < sh:
                iy1 += 1
            if iy1 <= iy0 or iy0 < 0:
                continue
            wy = np.ones(iy1 - iy0)
            if iy0 + 1 != iy1:
                wy[0] = 1 - (y0 - iy0)
                wy[-1] = 1 - (iy1 - y1)
            for j in range(dw):
                x0, x1 = max(kx * j + ox, 0.0), min(kx * (j + 1) + ox, float(sw))
                ix0, ix1 = int(np.floor(x0 + 1e-10)), int(np.ceil(x1 - 1e-10))
                if ix0 == ix1 and ix1 < sw:
                    ix1 += 1
                if ix1 <= ix0 or ix0 < 0:
                    continue
                wx = np.ones(ix1 - ix0)
                if ix0 + 1 != ix1:
                    wx[0] = 1 - (x0 - ix0)
                    wx[-1] = 1 - (ix1 - x1)
                if resampling in ('mode', 'med', 'q1', 'q3'):
                    # GWKAverageOrMode's rank-order branches (unweighted): quantile = element ceil(q * n - 1) of the sorted
                    # valid values; mode = the value whose running count first reaches the highest count (row-major scan)
                    vals = [srcd[a, b] for a in range(iy0, iy1) for b in range(ix0, ix1) if valid[a, b]]
                    if vals:
                        if resampling == 'mode':
                            counts, best, best_n = {}, None, 0
                            for v in vals:
                                counts[v] = counts.get(v, 0) + 1
                                if counts[v] > best_n:
                                    best, best_n = v, counts[v]
                            out[i, j] = best
                        else:
                            q = {'med': 0.5, 'q1': 0.25, 'q3': 0.75}[resampling]
                            out[i, j] = sorted(vals)[max(int(math.ceil(q * len(vals) - 1)), 0)]
                        got[i, j] = True
                    continue
                tot = wsum = 0.0
                for a in range(iy0, iy1):      # row-major accumulation order, as GDAL
                    for b in range(ix0, ix1):
                        if valid[a, b]:
                            wgt = wx[b - ix0] * wy[a - iy0]
                            v = srcd[a, b]
                            if resampling == 'max':
                                tot = max(tot, v) if wsum > 0 else v
                            elif resampling == 'min':
                                tot = min(tot, v) if wsum > 0 else v
                            elif resampling == 'rms':
                                tot += v * v * wgt
                            else:
                                tot += v * wgt
                            wsum += wgt
                if wsum > 0:
                    out[i, j] = tot / wsum if resampling == 'average' else (math.sqrt(tot / wsum) if resampling == 'rms' else tot)
                    got[i, j] = True
    elif resampling in ('bilinear', 'cubic_spline'):
        taps = (0, 1) if resampling == 'bilinear' else (-1, 0, 1, 2)
        for i in range(dh):
            sy = ky * (i + 0.5) + oy
            cy = int(np.floor(sy + 1e-10))
            iy = int(np.floor(sy - 0.5))
            dy = sy - 0.5 - iy
            wys = (1 - dy, dy) if resampling == 'bilinear' else _bspline_weights(dy)
            for j in range(dw):
                sx = kx * (j + 0.5) + ox
                cx = int(np.floor(sx + 1e-10))
                if cx < 0 or cx >= sw or cy < 0 or cy >= sh or not valid[cy, cx]:
                    continue   # the source pixel under the destination centre must be valid
                ix = int(np.floor(sx - 0.5))
                dx = sx - 0.5 - ix
                wxs = (1 - dx, dx) if resampling == 'bilinear' else _bspline_weights(dx)
                acc = wacc = 0.0
                for tj, wyv in zip(taps, wys):
                    a = iy + tj
                    if a < 0 or a >= sh:
                        continue
                    for ti, wxv in zip(taps, wxs):
                        b = ix + ti
                        if b < 0 or b >= sw or not valid[a, b]:
                            continue
                        wgt = float(wxv) * float(wyv)
                        acc += srcd[a, b] * wgt
                        wacc += wgt
                if wacc < 1e-6:
                    continue
                out[i, j] = acc / wacc if (wacc < 0.99999 or wacc > 1.00001) else acc
                got[i, j] = True
    else:
        raise NotImplementedError(resampling)
    res = out.astype(dtype)
    if dst_nodata is not None:
        res[~got] = dst_nodata
    return res


# ----------------------------------------------------------------------------------------------------------------------
# synthetic workload generator -- shared by golden generation, tests and the CPU-baseline sample
def synth_pair(h: int, w: int, seed: int = 0, nodata_variant: str = 'none', dn_like: bool = False):
    """
    ``src ~ U[0.05, 1)``, ``ref = f32(g(x,y)*src + o(y) + N(0, 0.01))`` with smooth gain/offset fields.
    nodata_variant: 'none' | 'frame+holes' (3-px NaN frame + 0.1 % random NaN pixels, independently in src & ref).
    """
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
    if dn_like:
        src = rng.normal(1000., 150., (h, w)).astype(F32)
        noise = rng.normal(0., 5., (h, w))
    else:
        src = rng.uniform(0.05, 1.0, (h, w)).astype(F32)
        noise = rng.normal(0., 0.01, (h, w))
    g = 1.2 + 0.3 * np.sin(xx / 97.) * np.cos(yy / 131.)
    o = 0.05 * (1 + 0.5 * np.sin(yy / 211.))
    if dn_like:
        o = o * 1000.
    ref = (g * src.astype(np.float64) + o + noise).astype(F32)
    if nodata_variant == 'frame+holes':
        for a in (src, ref):
            a[:3, :] = np.nan
            a[-3:, :] = np.nan
            a[:, :3] = np.nan
            a[:, -3:] = np.nan
            holes = rng.random((h, w)) < 0.001
            a[holes] = np.nan
    elif nodata_variant != 'none':
        raise ValueError(nodata_variant)
    return src, ref
