// hk_resample.hip -- re-sampling between same-CRS, north-up, axis-aligned grids: RasterArray.reproject
// (homonim/raster_array.py:526-578 -> rasterio.warp.reproject -> GDAL warp) as RefSpaceModel / SrcSpaceModel use it
// (homonim/kernel_model.py:397,480,491,497,520).
//
// GDAL is not part of /root/reference and not installed here: its published warp kernels (gdal/alg/gdalwarpkernel.cpp)
// are RESTATED, parity with GDAL itself unpinned; the arithmetic below is, operation for operation, the one of
// oracle/oracle_np.py::reproject, which the tests hold it to bit for bit.
//   mapping : src_col = kx * dst_col + ox, src_row = ky * dst_row + oy on continuous coordinates (integers = pixel edges)
//   0 nearest      : source pixel containing the destination centre (floor(x + 1e-10))
//   5 average      : weighted mean of the valid source pixels under the destination pixel's footprint
//   1 bilinear / 3 cubic_spline : centre pixel must be valid; separable 2 / 4-tap (cubic B-spline) kernel, invalid or
//                    outside taps skipped, renormalised by the accumulated weight (up-sampling: the fast path below)
//   1 bilinear / 2 cubic / 3 cubic_spline / 4 lanczos, any scale (resample_conv_kernel): GDAL's GWKResample -- taps
//                    i in [1 - R', R'] per axis with R' = ceil(R / scale) when the axis is down-sampled (scale =
//                    min(1, 1 / k) < 1), else R (1, 2, 2, 3); weight f((i - delta) * scale); always renormalised
//   8 max / 9 min / 13 sum / 14 rms : over the source pixels of the destination pixel's footprint (the window of
//                    `average`); sum and rms weight the edge pixels by their overlap like average
// One thread per destination pixel (gather); float64 accumulation, float32 result.
#include "hk_kernels.h"

namespace hk {

struct ResampleArgs {
    const float* src;
    float* dst;
    long long src_stride, src_band_stride, dst_stride, dst_band_stride;
    int sh, sw, dh, dw;
    int nd_mode;
    float nodata;
    float dst_fill;  // value of destination pixels that receive nothing
    double kx, ox, ky, oy;
};

__device__ __forceinline__ bool rs_valid(float v, int mode, float nodata) {
    return mode == 0 ? true : (mode == 1 ? !(v != v) : !(v == nodata));
}

__device__ __forceinline__ void bspline4(double d, double (&w)[4]) {
    const double a = 1.0 - d, b = 2.0 - d, c = 3.0 - d;
    w[0] = a * a * a / 6.0;
    w[1] = (b * b * b - 4.0 * (a * a * a)) / 6.0;
    w[2] = (c * c * c - 4.0 * (b * b * b) + 6.0 * (a * a * a)) / 6.0;
    w[3] = d * d * d / 6.0;
}

template <int MODE>
__global__ void __launch_bounds__(256) resample_kernel(const ResampleArgs a) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    const int i = blockIdx.y;
    if (j >= a.dw) return;
    const float* __restrict__ sp = a.src + (long long)blockIdx.z * a.src_band_stride;
    float* __restrict__ dp = a.dst + (long long)blockIdx.z * a.dst_band_stride;
    double result = 0.0;
    bool got = false;
    if constexpr (MODE == 0) {
        const long long cx = (long long)floor(a.kx * ((double)j + 0.5) + a.ox + 1e-10);
        const long long cy = (long long)floor(a.ky * ((double)i + 0.5) + a.oy + 1e-10);
        if (cx >= 0 && cx < a.sw && cy >= 0 && cy < a.sh) {
            const float v = sp[cy * a.src_stride + cx];
            if (rs_valid(v, a.nd_mode, a.nodata)) result = (double)v, got = true;
        }
    } else if constexpr (MODE == 5 || MODE == 8 || MODE == 9 || MODE == 13 || MODE == 14) {
        const double y0 = fmax(a.ky * (double)i + a.oy, 0.0), y1 = fmin(a.ky * (double)(i + 1) + a.oy, (double)a.sh);
        const double x0 = fmax(a.kx * (double)j + a.ox, 0.0), x1 = fmin(a.kx * (double)(j + 1) + a.ox, (double)a.sw);
        int iy0 = (int)floor(y0 + 1e-10), iy1 = (int)ceil(y1 - 1e-10);
        int ix0 = (int)floor(x0 + 1e-10), ix1 = (int)ceil(x1 - 1e-10);
        if (iy0 == iy1 && iy1 < a.sh) ++iy1;
        if (ix0 == ix1 && ix1 < a.sw) ++ix1;
        if (iy1 > iy0 && iy0 >= 0 && ix1 > ix0 && ix0 >= 0) {
            double tot = 0.0, wsum = 0.0;
            for (int yy = iy0; yy < iy1; ++yy) {
                double wy = 1.0;
                if (iy0 + 1 != iy1) wy = yy == iy0 ? 1.0 - (y0 - (double)iy0) : (yy == iy1 - 1 ? 1.0 - ((double)iy1 - y1) : 1.0);
                for (int xx = ix0; xx < ix1; ++xx) {
                    const float v = sp[(long long)yy * a.src_stride + xx];
                    if (!rs_valid(v, a.nd_mode, a.nodata)) continue;
                    double wx = 1.0;
                    if (ix0 + 1 != ix1) wx = xx == ix0 ? 1.0 - (x0 - (double)ix0) : (xx == ix1 - 1 ? 1.0 - ((double)ix1 - x1) : 1.0);
                    const double wgt = wx * wy;
                    if constexpr (MODE == 8) {
                        tot = wsum > 0.0 ? fmax(tot, (double)v) : (double)v;
                    } else if constexpr (MODE == 9) {
                        tot = wsum > 0.0 ? fmin(tot, (double)v) : (double)v;
                    } else if constexpr (MODE == 14) {
                        tot += (double)v * (double)v * wgt;
                    } else {
                        tot += (double)v * wgt;
                    }
                    wsum += wgt;
                }
            }
            if (wsum > 0.0) {
                result = (MODE == 5) ? tot / wsum : ((MODE == 14) ? sqrt(tot / wsum) : tot);
                got = true;
            }
        }
    } else if constexpr (MODE == 6 || MODE == 10 || MODE == 11 || MODE == 12) {
        // GWKAverageOrMode's rank-order branches over the same footprint as `average` (no weights): med / q1 / q3 = element
        // ceil(q * n - 1) of the sorted valid values; mode = the value whose running count first reaches the highest
        // count, in row-major scan order.  No per-thread storage: the footprint (a few dozen pixels, cache-resident) is
        // scanned once per candidate.
        const double y0 = fmax(a.ky * (double)i + a.oy, 0.0), y1 = fmin(a.ky * (double)(i + 1) + a.oy, (double)a.sh);
        const double x0 = fmax(a.kx * (double)j + a.ox, 0.0), x1 = fmin(a.kx * (double)(j + 1) + a.ox, (double)a.sw);
        int iy0 = (int)floor(y0 + 1e-10), iy1 = (int)ceil(y1 - 1e-10);
        int ix0 = (int)floor(x0 + 1e-10), ix1 = (int)ceil(x1 - 1e-10);
        if (iy0 == iy1 && iy1 < a.sh) ++iy1;
        if (ix0 == ix1 && ix1 < a.sw) ++ix1;
        if (iy1 > iy0 && iy0 >= 0 && ix1 > ix0 && ix0 >= 0) {
            int n = 0;
            for (int yy = iy0; yy < iy1; ++yy)
                for (int xx = ix0; xx < ix1; ++xx) n += rs_valid(sp[(long long)yy * a.src_stride + xx], a.nd_mode, a.nodata) ? 1 : 0;
            if (n > 0) {
                constexpr double q = MODE == 10 ? 0.5 : (MODE == 11 ? 0.25 : 0.75);
                int want = (int)ceil(q * (double)n - 1.0);
                want = want < 0 ? 0 : want;
                int best_cnt = 0, best_last = 0;
                for (int yc = iy0; yc < iy1 && !(MODE != 6 && got); ++yc) {
                    for (int xc = ix0; xc < ix1; ++xc) {
                        const float c = sp[(long long)yc * a.src_stride + xc];
                        if (!rs_valid(c, a.nd_mode, a.nodata)) continue;
                        int less = 0, equal = 0, last = 0, pos = 0;
                        for (int yy = iy0; yy < iy1; ++yy)
                            for (int xx = ix0; xx < ix1; ++xx, ++pos) {
                                const float v = sp[(long long)yy * a.src_stride + xx];
                                if (!rs_valid(v, a.nd_mode, a.nodata)) continue;
                                less += v < c ? 1 : 0;
                                if (v == c) ++equal, last = pos;
                            }
                        if constexpr (MODE == 6) {
                            // the value that reaches the highest count first = most occurrences, then earliest last occurrence
                            if (equal > best_cnt || (equal == best_cnt && last < best_last))
                                best_cnt = equal, best_last = last, result = (double)c, got = true;
                        } else if (less <= want && want < less + equal) {
                            result = (double)c, got = true;
                            break;
                        }
                    }
                }
            }
        }
    } else {
        constexpr int NT = MODE == 1 ? 2 : 4, T0 = MODE == 1 ? 0 : -1;
        const double sy = a.ky * ((double)i + 0.5) + a.oy, sx = a.kx * ((double)j + 0.5) + a.ox;
        const long long cy = (long long)floor(sy + 1e-10), cx = (long long)floor(sx + 1e-10);
        if (cx >= 0 && cx < a.sw && cy >= 0 && cy < a.sh && rs_valid(sp[cy * a.src_stride + cx], a.nd_mode, a.nodata)) {
            const int iy = (int)floor(sy - 0.5), ix = (int)floor(sx - 0.5);
            const double dy = sy - 0.5 - (double)iy, dx = sx - 0.5 - (double)ix;
            double wys[4], wxs[4];
            if constexpr (MODE == 1) {
                wys[0] = 1.0 - dy, wys[1] = dy, wxs[0] = 1.0 - dx, wxs[1] = dx;
            } else {
                bspline4(dy, wys);
                bspline4(dx, wxs);
            }
            double acc = 0.0, wacc = 0.0;
            for (int tj = 0; tj < NT; ++tj) {
                const int yy = iy + T0 + tj;
                if (yy < 0 || yy >= a.sh) continue;
                for (int ti = 0; ti < NT; ++ti) {
                    const int xx = ix + T0 + ti;
                    if (xx < 0 || xx >= a.sw) continue;
                    const float v = sp[(long long)yy * a.src_stride + xx];
                    if (!rs_valid(v, a.nd_mode, a.nodata)) continue;
                    const double wgt = wxs[ti] * wys[tj];
                    acc += (double)v * wgt;
                    wacc += wgt;
                }
            }
            if (!(wacc < 1e-6)) {
                result = (wacc < 0.99999 || wacc > 1.00001) ? acc / wacc : acc;
                got = true;
            }
        }
    }
    dp[(long long)i * a.dst_stride + j] = got ? (float)result : a.dst_fill;
}

// GDAL's re-sampling kernels as functions of the (scaled) distance: GWKBilinear / GWKCubic (a = -0.5) / GWKBSpline /
// GWKLanczosSinc (radius 3)
template <int KIND>
__device__ __forceinline__ double conv_weight(double x) {
    const double ax = fabs(x);
    if constexpr (KIND == 1) {
        return ax <= 1.0 ? 1.0 - ax : 0.0;
    } else if constexpr (KIND == 2) {
        const double x2 = ax * ax;
        if (ax <= 1.0) return x2 * (1.5 * ax - 2.5) + 1.0;
        if (ax <= 2.0) return x2 * (-0.5 * ax + 2.5) - 4.0 * ax + 2.0;
        return 0.0;
    } else if constexpr (KIND == 3) {
        if (ax > 2.0) return 0.0;
        const double xp2 = x + 2.0, xp1 = x + 1.0, xm1 = x - 1.0;
        const double a = xp2 > 0.0 ? xp2 * xp2 * xp2 : 0.0, b = xp1 > 0.0 ? xp1 * xp1 * xp1 : 0.0;
        const double c = x > 0.0 ? x * x * x : 0.0, d = xm1 > 0.0 ? xm1 * xm1 * xm1 : 0.0;
        return (a - 4.0 * b + 6.0 * c - 4.0 * d) / 6.0;
    } else {
        if (ax >= 3.0) return 0.0;
        if (x == 0.0) return 1.0;
        const double pi = 3.14159265358979323846, px = pi * x, px3 = px / 3.0;
        return sin(px) * sin(px3) / (px * px3);
    }
}

// GWKResample for any scale: bilinear (1) / cubic (2) / cubic_spline (3) / lanczos (4)
template <int KIND>
__global__ void __launch_bounds__(256) resample_conv_kernel(const ResampleArgs a) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    const int i = blockIdx.y;
    if (j >= a.dw) return;
    const float* __restrict__ sp = a.src + (long long)blockIdx.z * a.src_band_stride;
    float* __restrict__ dp = a.dst + (long long)blockIdx.z * a.dst_band_stride;
    constexpr int R = KIND == 1 ? 1 : (KIND == 4 ? 3 : 2);
    const double xs = a.kx > 1.0 ? 1.0 / a.kx : 1.0, ys = a.ky > 1.0 ? 1.0 / a.ky : 1.0;
    const int rx = xs < 1.0 ? (int)ceil((double)R / xs) : R, ry = ys < 1.0 ? (int)ceil((double)R / ys) : R;
    const double sy = a.ky * ((double)i + 0.5) + a.oy, sx = a.kx * ((double)j + 0.5) + a.ox;
    const long long cy = (long long)floor(sy + 1e-10), cx = (long long)floor(sx + 1e-10);
    double result = 0.0;
    bool got = false;
    if (cx >= 0 && cx < a.sw && cy >= 0 && cy < a.sh && rs_valid(sp[cy * a.src_stride + cx], a.nd_mode, a.nodata)) {
        const int iy = (int)floor(sy - 0.5), ix = (int)floor(sx - 0.5);
        const double dy = sy - 0.5 - (double)iy, dx = sx - 0.5 - (double)ix;
        double acc = 0.0, wacc = 0.0;
        for (int tj = 1 - ry; tj <= ry; ++tj) {
            const int yy = iy + tj;
            if (yy < 0 || yy >= a.sh) continue;
            const double wy = conv_weight<KIND>(((double)tj - dy) * ys);
            if (wy == 0.0) continue;
            for (int ti = 1 - rx; ti <= rx; ++ti) {
                const int xx = ix + ti;
                if (xx < 0 || xx >= a.sw) continue;
                const float v = sp[(long long)yy * a.src_stride + xx];
                if (!rs_valid(v, a.nd_mode, a.nodata)) continue;
                const double wgt = conv_weight<KIND>(((double)ti - dx) * xs) * wy;
                acc += (double)v * wgt;
                wacc += wgt;
            }
        }
        if (!(fabs(wacc) < 1e-6)) result = acc / wacc, got = true;
    }
    dp[(long long)i * a.dst_stride + j] = got ? (float)result : a.dst_fill;
}

// valid(src) as a float32 0/1 plane: RasterArray.mask_ra (raster_array.py:320-327) before it is re-projected
__global__ void __launch_bounds__(256) valid_plane_kernel(const float* __restrict__ in, long long in_stride, int nd_mode,
                                                          float nodata, float* __restrict__ out, long long out_stride,
                                                          int height, int width) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    if (x >= width) return;
    for (int y = blockIdx.y; y < height; y += gridDim.y)
        out[(long long)y * out_stride + x] = rs_valid(in[(long long)y * in_stride + x], nd_mode, nodata) ? 1.f : 0.f;
}

// RefSpaceModel.apply after the parameters were brought to the source grid (kernel_model.py:493-503): parameters are
// masked with the (nearest re-projected) full-coverage mask, or with the source mask, then gain * src + offset.
__global__ void __launch_bounds__(256) apply_space_kernel(const float* __restrict__ src, long long src_stride, int nd_mode,
                                                          float nodata, const float* __restrict__ gain,
                                                          const float* __restrict__ offset, long long par_stride,
                                                          const float* __restrict__ keep, float* __restrict__ out,
                                                          long long out_stride, int height, int width) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    if (x >= width) return;
    const float nan = __int_as_float(0x7fc00000);
    for (int y = blockIdx.y; y < height; y += gridDim.y) {
        const float s = src[(long long)y * src_stride + x];
        const long long pi = (long long)y * par_stride + x;
        // param_us_ra.mask = mask_us (bool of the nearest-resampled mask, :498) or src_ra.mask (:500)
        const bool on = keep ? keep[pi] != 0.f : rs_valid(s, nd_mode, nodata);
        const float g = on ? gain[pi] : nan, o = on ? offset[pi] : nan;
        out[(long long)y * out_stride + x] = __fadd_rn(__fmul_rn(g, s), o);
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// RefSpaceModel.apply fused (kernel_model.py:484-503): the up-sampled gain / offset never exist as full-resolution
// planes.  Per destination pixel the bilinear / cubic-spline values of BOTH parameter planes are formed exactly as
// resample_kernel<MODE> does (same weights, same tap order, same renormalisation rule), masked like apply_space_kernel and
// applied.  The work that does not depend on the pixel is hoisted: the row geometry and weights come from a table made
// by row_table_kernel (one entry per destination row), the column geometry and weights are computed once per thread,
// which then walks down its column.
constexpr int UP_ROWS = 16;  // destination rows per thread of upsample_apply_kernel (amortises the column weights)
struct RowTab {
    double w[4];
    int iy;       // first tap row - T0
    int centre;   // source row under the destination centre, or -1 when outside
    int yy[4];    // tap rows clamped into the plane (always loadable)
    int all_in;   // every tap row lies inside the plane
};

template <int MODE>
__global__ void __launch_bounds__(256) row_table_kernel(RowTab* __restrict__ tab, int dh, int sh, double ky, double oy) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= dh) return;
    const double sy = ky * ((double)i + 0.5) + oy;
    const long long cy = (long long)floor(sy + 1e-10);
    RowTab t;
    t.iy = (int)floor(sy - 0.5);
    t.centre = (cy >= 0 && cy < sh) ? (int)cy : -1;
    const double dy = sy - 0.5 - (double)t.iy;
    if constexpr (MODE == 1) {
        t.w[0] = 1.0 - dy, t.w[1] = dy, t.w[2] = t.w[3] = 0.0;
    } else {
        bspline4(dy, t.w);
    }
    constexpr int NT = MODE == 1 ? 2 : 4, T0 = MODE == 1 ? 0 : -1;
    t.all_in = 1;
    for (int tj = 0; tj < 4; ++tj) {
        const int yy = t.iy + T0 + (tj < NT ? tj : NT - 1);
        if (yy < 0 || yy >= sh) t.all_in = 0;
        t.yy[tj] = min(max(yy, 0), sh - 1);
    }
    tab[i] = t;
}

struct UpApplyArgs {
    const float* src;     // full-resolution source (destination grid)
    long long src_stride;
    int nd_mode;
    float nodata;
    const float* gain;    // coarse parameter planes, NaN = nodata
    const float* offset;
    long long par_stride;
    int ph, pw;
    const float* keep;    // nullable full-resolution 0/1 plane (mask_partial)
    long long keep_stride;
    float* out;
    long long out_stride;
    int height, width;
    double kx, ox;
    const RowTab* rows;
};

template <int MODE>
__global__ void __launch_bounds__(256) upsample_apply_kernel(const UpApplyArgs a) {
    constexpr int NT = MODE == 1 ? 2 : 4, T0 = MODE == 1 ? 0 : -1;
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= a.width) return;
    const float nan = __int_as_float(0x7fc00000);
    // column geometry and weights: once per thread
    const double sx = a.kx * ((double)j + 0.5) + a.ox;
    const long long cx = (long long)floor(sx + 1e-10);
    const bool cx_ok = cx >= 0 && cx < a.pw;
    const int ix = (int)floor(sx - 0.5);
    const double dx = sx - 0.5 - (double)ix;
    double wxs[4];
    if constexpr (MODE == 1) {
        wxs[0] = 1.0 - dx, wxs[1] = dx, wxs[2] = wxs[3] = 0.0;
    } else {
        bspline4(dx, wxs);
    }
    // first tap column, clamped so that NT consecutive columns are always loadable (straight-line path below)
    const bool col_all_in = ix + T0 >= 0 && ix + T0 + NT <= a.pw;
    const int xbase = min(max(ix + T0, 0), max(a.pw - NT, 0));
    // consecutive destination rows per block: neighbouring rows read the same parameter rows (L1 / L2 reuse)
    const int i_end = min(a.height, ((int)blockIdx.y + 1) * UP_ROWS);
    for (int i = blockIdx.y * UP_ROWS; i < i_end; ++i) {
        const RowTab t = a.rows[i];  // wave-uniform
        const float s = a.src[(long long)i * a.src_stride + j];
        const bool on = a.keep ? a.keep[(long long)i * a.keep_stride + j] != 0.f : rs_valid(s, a.nd_mode, a.nodata);
        float par[2] = {nan, nan};
        const bool need = on && cx_ok && t.centre >= 0;
        // Straight-line path (the interior of the raster): every tap of every lane that needs a value is inside the plane
        // and not NaN -> all 2 x NT x NT loads go out together, no per-tap control flow, and the sums run over the same
        // taps in the same order as below (the weight sum does not depend on the plane).  Otherwise: the general path.
        float tg[NT][NT], to[NT][NT];
        float chk = 0.f;
#pragma unroll
        for (int tj = 0; tj < NT; ++tj) {
            // the NT taps of a row are consecutive columns when they are all inside (the only case that uses them):
            // one (possibly unaligned) NT-float load per row and plane from a base clamped into the plane
            const int off = t.yy[tj] * (int)a.par_stride + xbase;
            if constexpr (NT == 4) {
                const float4 vg = *reinterpret_cast<const float4*>(a.gain + off);
                const float4 vo = *reinterpret_cast<const float4*>(a.offset + off);
                tg[tj][0] = vg.x, tg[tj][1] = vg.y, tg[tj][2] = vg.z, tg[tj][3] = vg.w;
                to[tj][0] = vo.x, to[tj][1] = vo.y, to[tj][2] = vo.z, to[tj][3] = vo.w;
            } else {
                const float2 vg = *reinterpret_cast<const float2*>(a.gain + off);
                const float2 vo = *reinterpret_cast<const float2*>(a.offset + off);
                tg[tj][0] = vg.x, tg[tj][1] = vg.y, to[tj][0] = vo.x, to[tj][1] = vo.y;
            }
#pragma unroll
            for (int ti = 0; ti < NT; ++ti) chk += tg[tj][ti] + to[tj][ti];  // NaN anywhere (or inf - inf) makes chk NaN
        }
        const bool straight = !need || (t.all_in && col_all_in && chk == chk);
        if (__all((int)straight)) {
            double wgt[NT][NT], wacc = 0.0, accg = 0.0, acco = 0.0;
#pragma unroll
            for (int tj = 0; tj < NT; ++tj) {
#pragma unroll
                for (int ti = 0; ti < NT; ++ti) {
                    wgt[tj][ti] = wxs[ti] * t.w[tj];
                    accg += (double)tg[tj][ti] * wgt[tj][ti];
                    acco += (double)to[tj][ti] * wgt[tj][ti];
                    wacc += wgt[tj][ti];
                }
            }
            const bool renorm = wacc < 0.99999 || wacc > 1.00001;
            if (__any((int)(need && renorm))) {  // wave-uniform: interior weights sum to 1 within the tolerance
                accg = renorm ? accg / wacc : accg;
                acco = renorm ? acco / wacc : acco;
            }
            if (need && !(wacc < 1e-6)) par[0] = (float)accg, par[1] = (float)acco;
        } else if (need) {
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const float* __restrict__ pp = b ? a.offset : a.gain;
                if (pp[(long long)t.centre * a.par_stride + cx] != pp[(long long)t.centre * a.par_stride + cx]) continue;  // NaN centre
                double acc = 0.0, wacc = 0.0;
#pragma unroll
                for (int tj = 0; tj < NT; ++tj) {
                    const int yy = t.iy + T0 + tj;
                    if (yy < 0 || yy >= a.ph) continue;
#pragma unroll
                    for (int ti = 0; ti < NT; ++ti) {
                        const int xx = ix + T0 + ti;
                        if (xx < 0 || xx >= a.pw) continue;
                        const float v = pp[(long long)yy * a.par_stride + xx];
                        if (v != v) continue;
                        const double wgt = wxs[ti] * t.w[tj];
                        acc += (double)v * wgt;
                        wacc += wgt;
                    }
                }
                if (!(wacc < 1e-6)) par[b] = (float)((wacc < 0.99999 || wacc > 1.00001) ? acc / wacc : acc);
            }
        }
        a.out[(long long)i * a.out_stride + j] = __fadd_rn(__fmul_rn(par[0], s), par[1]);
    }
}

size_t upsample_apply_workspace_bytes(int height) { return (size_t)height * sizeof(RowTab); }

// mode: 1 bilinear, 3 cubic_spline (anything else: hipErrorInvalidValue -- the caller keeps the unfused path)
hipError_t launch_upsample_apply(int mode, const float* src, long long src_stride, int nd_mode, float nodata,
                                 const float* gain, const float* offset, long long par_stride, int ph, int pw,
                                 const float* keep, long long keep_stride, float* out, long long out_stride, int height,
                                 int width, double kx, double ox, double ky, double oy, void* workspace, hipStream_t stream) {
    if (mode != 1 && mode != 3) return hipErrorInvalidValue;
    if ((long long)ph * par_stride >= 0x7fffffffLL) return hipErrorInvalidValue;  // 32-bit tap offsets
    if (pw < 4) return hipErrorInvalidValue;                                       // row loads of 4 consecutive taps
    RowTab* tab = static_cast<RowTab*>(workspace);
    UpApplyArgs a;
    a.src = src, a.src_stride = src_stride, a.nd_mode = nd_mode, a.nodata = nodata, a.gain = gain, a.offset = offset;
    a.par_stride = par_stride, a.ph = ph, a.pw = pw, a.keep = keep, a.keep_stride = keep_stride, a.out = out;
    a.out_stride = out_stride, a.height = height, a.width = width, a.kx = kx, a.ox = ox, a.rows = tab;
    const dim3 tgrid((height + 255) / 256), grid((width + 255) / 256, (height + UP_ROWS - 1) / UP_ROWS), block(256);
    if (mode == 1) {
        HK_LAUNCH(row_table_kernel<1>, tgrid, block, 0, stream, tab, height, ph, ky, oy);
        HK_LAUNCH(upsample_apply_kernel<1>, grid, block, 0, stream, a);
    } else {
        HK_LAUNCH(row_table_kernel<3>, tgrid, block, 0, stream, tab, height, ph, ky, oy);
        HK_LAUNCH(upsample_apply_kernel<3>, grid, block, 0, stream, a);
    }
    return hipGetLastError();
}

hipError_t launch_valid_plane(const float* in, long long in_stride, int nd_mode, float nodata, float* out,
                              long long out_stride, int height, int width, hipStream_t stream) {
    HK_LAUNCH(valid_plane_kernel, dim3((width + 255) / 256, height < 1024 ? height : 1024), dim3(256), 0, stream,
                       in, in_stride, nd_mode, nodata, out, out_stride, height, width);
    return hipGetLastError();
}

hipError_t launch_apply_space(const float* src, long long src_stride, int nd_mode, float nodata, const float* gain,
                              const float* offset, long long par_stride, const float* keep, float* out,
                              long long out_stride, int height, int width, hipStream_t stream) {
    HK_LAUNCH(apply_space_kernel, dim3((width + 255) / 256, height < 1024 ? height : 1024), dim3(256), 0, stream,
                       src, src_stride, nd_mode, nodata, gain, offset, par_stride, keep, out, out_stride, height, width);
    return hipGetLastError();
}

hipError_t launch_resample(int mode, const float* src, long long src_stride, long long src_band_stride, int sh, int sw,
                           int n_bands, int nd_mode, float nodata, double kx, double ox, double ky, double oy, float* dst,
                           long long dst_stride, long long dst_band_stride, int dh, int dw, float dst_fill,
                           hipStream_t stream) {
    ResampleArgs a;
    a.src = src, a.dst = dst, a.src_stride = src_stride, a.src_band_stride = src_band_stride, a.dst_stride = dst_stride;
    a.dst_band_stride = dst_band_stride, a.sh = sh, a.sw = sw, a.dh = dh, a.dw = dw, a.nd_mode = nd_mode, a.nodata = nodata;
    a.dst_fill = dst_fill, a.kx = kx, a.ox = ox, a.ky = ky, a.oy = oy;
    const dim3 grid((dw + 255) / 256, dh, n_bands), block(256);
    const bool stretched = kx > 1.0 + 1e-9 || ky > 1.0 + 1e-9;  // an axis is down-sampled: the kernel support scales with it
    switch (mode) {
        case 0: HK_LAUNCH(resample_kernel<0>, grid, block, 0, stream, a); break;
        case 1:
            if (stretched) HK_LAUNCH(resample_conv_kernel<1>, grid, block, 0, stream, a);
            else HK_LAUNCH(resample_kernel<1>, grid, block, 0, stream, a);
            break;
        case 2: HK_LAUNCH(resample_conv_kernel<2>, grid, block, 0, stream, a); break;
        case 3:
            if (stretched) HK_LAUNCH(resample_conv_kernel<3>, grid, block, 0, stream, a);
            else HK_LAUNCH(resample_kernel<3>, grid, block, 0, stream, a);
            break;
        case 4: HK_LAUNCH(resample_conv_kernel<4>, grid, block, 0, stream, a); break;
        case 5: HK_LAUNCH(resample_kernel<5>, grid, block, 0, stream, a); break;
        case 6: HK_LAUNCH(resample_kernel<6>, grid, block, 0, stream, a); break;
        case 10: HK_LAUNCH(resample_kernel<10>, grid, block, 0, stream, a); break;
        case 11: HK_LAUNCH(resample_kernel<11>, grid, block, 0, stream, a); break;
        case 12: HK_LAUNCH(resample_kernel<12>, grid, block, 0, stream, a); break;
        case 8: HK_LAUNCH(resample_kernel<8>, grid, block, 0, stream, a); break;
        case 9: HK_LAUNCH(resample_kernel<9>, grid, block, 0, stream, a); break;
        case 13: HK_LAUNCH(resample_kernel<13>, grid, block, 0, stream, a); break;
        case 14: HK_LAUNCH(resample_kernel<14>, grid, block, 0, stream, a); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

}  // namespace hk
