/*
 * ORACLE -- TEST INFRASTRUCTURE ONLY.  Not part of the product path, never linked or loaded by homonim_amd.
 *
 * Plain-C restatement of the homonim kernel-model fit/apply hot path (reference: leftfield-geospatial/homonim
 * v0.4.3, homonim/kernel_model.py).  It is the on-box checker of the HIP kernels and bench.py's CPU baseline
 * (cpu_baseline.kind = "port").  Each function cites the reference lines it follows; the arithmetic (types,
 * operation order, summation order) is identical to oracle/oracle_np.py, which is pinned bit-for-bit against
 * golden vectors produced by the reference itself and against the reference's PARAM GeoTIFF -- and this file is
 * pinned bit-for-bit against oracle_np (tests/test_oracle_c.py).
 *
 * OpenCV (cv.boxFilter / cv.sqrBoxFilter, normalize=False, BORDER_CONSTANT), which the reference calls at
 * kernel_model.py:167-175,184,257-258,332-341, is not part of /root/reference: its published algorithm is restated
 * as a zero-border, centre-anchored window sum accumulated in float64; boxFilter returns the input depth,
 * sqrBoxFilter returns float64 (its ddepth=-1 rule) -- see oracle_np.box_sum.
 *
 * Build: gcc -O2 -std=c11 -fPIC -shared -fopenmp -ffp-contract=off -fno-fast-math hk_oracle.c -lm
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <limits.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

enum { MODEL_GAIN = 0, MODEL_GAIN_BLK_OFFSET = 1, MODEL_GAIN_OFFSET = 2 };
enum { ND_NONE = 0, ND_NAN = 1, ND_VALUE = 2 };

/* ~utils.nan_equals(v, nodata) (homonim/utils.py:54-56; raster_array.py:298-308) */
static inline int px_valid(float v, int mode, float nodata) {
    if (mode == ND_NONE) return 1;
    if (mode == ND_NAN) return !isnan(v);
    return !(v == nodata);
}

/* Row-streaming window sums.  For every output row the kw-tap horizontal sums of the kh contributing rows are kept in
 * a ring; the vertical sum adds them top to bottom -- the same order as oracle_np.box_sum (columns left to right,
 * then rows top to bottom), so float64 results are bit-identical to it. */
typedef struct {
    int kh, kw, rh, rw, W;
    int nq;          /* quantities held */
    double* ring;    /* [kh][nq][W] horizontal sums */
    double* zrow;    /* [nq][W + 2 rw] zero-padded per-pixel terms of the row being inserted */
} WinState;

enum { Q_S = 0, Q_R = 1, Q_P = 2, Q_N = 3, Q_S2 = 4, Q_R2 = 5, NQ = 6 };

/* terms of one input row: masked source/reference (float64 for the gain-blk-offset normalised source), their
 * float32-rounded product (or float64 product for gain-blk-offset), mask, and squares formed in float64 */
static void row_terms(const WinState* ws, int model, const float* src, const float* ref, int row_ok, int W,
                      int snd_mode, float snd, int rnd_mode, float rnd, double n0, double n1, unsigned char* mrow,
                      float* szrow) {
    const int pw = W + 2 * ws->rw;
    double* z = ws->zrow;
    memset(z, 0, sizeof(double) * (size_t)NQ * pw);
    for (int x = 0; x < W; ++x) {
        int m = 0;
        float s = 0.f, r = 0.f;
        double sd = 0.0;
        if (row_ok) {
            s = src[x];
            r = ref[x];
            m = px_valid(s, snd_mode, snd) && px_valid(r, rnd_mode, rnd);
            if (model == MODEL_GAIN_BLK_OFFSET) {
                /* kernel_model.py:292-298: nodata -> nan, src*norm[0]+norm[1] in float64 (NumPy>=2), mask = ~isnan */
                const int ms = px_valid(s, snd_mode, snd);
                sd = ms ? ((double)s * n0) + n1 : NAN;
                m = !isnan(sd) && px_valid(r, rnd_mode, rnd);
            }
        }
        if (mrow) mrow[x] = (unsigned char)m;
        if (szrow) szrow[x] = m ? s : 0.f;
        if (!m) continue;   /* zero-filled: kernel_model.py:246-247,320-321 */
        const int px = x + ws->rw;
        const double dr = (double)r;
        if (model == MODEL_GAIN_BLK_OFFSET) {
            z[Q_S * pw + px] = sd;
            z[Q_P * pw + px] = sd * dr;          /* boxFilter(src64 * ref32): float64 product */
            z[Q_S2 * pw + px] = sd * sd;
        } else {
            const double ds = (double)s;
            z[Q_S * pw + px] = ds;
            z[Q_P * pw + px] = (double)(float)(s * r);   /* numpy rounds src*ref to float32 first (:175,:334) */
            z[Q_S2 * pw + px] = ds * ds;
        }
        z[Q_R * pw + px] = dr;
        z[Q_N * pw + px] = 1.0;
        z[Q_R2 * pw + px] = dr * dr;
    }
}

static void hsum_into_ring(const WinState* ws, int slot) {
    const int W = ws->W, pw = W + 2 * ws->rw, kw = ws->kw;
    for (int q = 0; q < NQ; ++q) {
        const double* z = ws->zrow + (size_t)q * pw;
        double* out = ws->ring + ((size_t)slot * NQ + q) * W;
        for (int x = 0; x < W; ++x) {
            double acc = 0.0;
            for (int dx = 0; dx < kw; ++dx) acc += z[x + dx];
            out[x] = acc;
        }
    }
}

/* (defined below) */
int hk_oracle_fill_nodata(float* image, const unsigned char* mask, int H, int W, double max_search_distance);

/*
 * KernelModel.fit (+ apply) on one band (kernel_model.py:411-463).
 *   params_out: n_param_bands x H x W or NULL; corr_out: H x W or NULL; norm: float64[2] for gain-blk-offset.
 *   with_r2 = find_r2 || (gain-offset && has_thresh)  (kernel_model.py:252,325)
 * Returns 0, or -1 on bad arguments / out of memory.
 */
int hk_oracle_fit_apply(int model, int kh, int kw, int find_r2, int has_thresh, float thresh, const float* src,
                        int snd_mode, float snd, const float* ref, int rnd_mode, float rnd, int H, int W,
                        const double* norm, float* params_out, int n_param_bands, float* corr_out,
                        uint64_t* fail_count, int n_threads) {
    if (kh < 1 || kw < 1 || !(kh & 1) || !(kw & 1) || H < 1 || W < 1 || !src || !ref) return -1;
    const int with_r2 = find_r2 || (model == MODEL_GAIN_OFFSET && has_thresh);
    if (params_out && n_param_bands != (with_r2 ? 3 : 2)) return -1;
    if (model == MODEL_GAIN_BLK_OFFSET && !norm) return -1;
    const double n0 = norm ? norm[0] : 0.0, n1 = norm ? norm[1] : 0.0;
    const int rh = kh / 2, rw = kw / 2;
    const size_t plane = (size_t)H * W;
    uint64_t fails = 0;
    int err = 0;
    /* in-paint branch (kernel_model.py:361-371) needs the parameters and three of the window sums afterwards */
    const int may_inpaint = model == MODEL_GAIN_OFFSET && has_thresh;
    float* own_params = NULL;
    float *keepS = NULL, *keepR = NULL, *keepN = NULL;
    if (may_inpaint) {
        if (!params_out) {
            own_params = (float*)malloc(sizeof(float) * 3 * plane);
            if (!own_params) return -1;
            params_out = own_params;
        }
        keepS = (float*)malloc(sizeof(float) * plane), keepR = (float*)malloc(sizeof(float) * plane);
        keepN = (float*)malloc(sizeof(float) * plane);
        if (!keepS || !keepR || !keepN) {
            free(own_params), free(keepS), free(keepR), free(keepN);
            return -1;
        }
    }
#ifdef _OPENMP
    if (n_threads < 1) n_threads = omp_get_max_threads();
#else
    n_threads = 1;
#endif
    if (n_threads > H) n_threads = H;

#pragma omp parallel num_threads(n_threads) reduction(+ : fails) reduction(| : err)
    {
#ifdef _OPENMP
        const int tid = omp_get_thread_num(), nt = omp_get_num_threads();
#else
        const int tid = 0, nt = 1;
#endif
        const int y_begin = (int)((int64_t)H * tid / nt), y_end = (int)((int64_t)H * (tid + 1) / nt);
        WinState ws;
        ws.kh = kh, ws.kw = kw, ws.rh = rh, ws.rw = rw, ws.W = W, ws.nq = NQ;
        ws.ring = (double*)malloc(sizeof(double) * (size_t)kh * NQ * W);
        ws.zrow = (double*)malloc(sizeof(double) * (size_t)NQ * (W + 2 * rw));
        unsigned char* mring = (unsigned char*)malloc((size_t)kh * W);
        float* sring = (float*)malloc(sizeof(float) * (size_t)kh * W);
        double* acc = (double*)malloc(sizeof(double) * (size_t)NQ * W);
        if (!ws.ring || !ws.zrow || !mring || !sring || !acc) {
            err = 1;
        } else if (y_begin < y_end) {
            /* prime the ring with rows y_begin-rh .. y_begin+rh-1 */
            for (int t = y_begin - rh; t < y_begin + rh; ++t) {
                const int slot = ((t % kh) + kh) % kh;
                const int ok = t >= 0 && t < H;
                row_terms(&ws, model, ok ? src + (size_t)t * W : NULL, ok ? ref + (size_t)t * W : NULL, ok, W, snd_mode,
                          snd, rnd_mode, rnd, n0, n1, mring + (size_t)slot * W, sring + (size_t)slot * W);
                hsum_into_ring(&ws, slot);
            }
            for (int y = y_begin; y < y_end; ++y) {
                const int t = y + rh, slot = ((t % kh) + kh) % kh;
                const int ok = t >= 0 && t < H;
                row_terms(&ws, model, ok ? src + (size_t)t * W : NULL, ok ? ref + (size_t)t * W : NULL, ok, W, snd_mode,
                          snd, rnd_mode, rnd, n0, n1, mring + (size_t)slot * W, sring + (size_t)slot * W);
                hsum_into_ring(&ws, slot);
                /* vertical sums, top row first */
                memset(acc, 0, sizeof(double) * (size_t)NQ * W);
                for (int dy = -rh; dy <= rh; ++dy) {
                    const int sl = (((y + dy) % kh) + kh) % kh;
                    for (int q = 0; q < NQ; ++q) {
                        const double* r_ = ws.ring + ((size_t)sl * NQ + q) * W;
                        double* a_ = acc + (size_t)q * W;
                        for (int x = 0; x < W; ++x) a_[x] += r_[x];
                    }
                }
                const int cslot = ((y % kh) + kh) % kh;
                const unsigned char* mc = mring + (size_t)cslot * W;
                const float* sc = sring + (size_t)cslot * W;
                for (int x = 0; x < W; ++x) {
                    float g = NAN, o = NAN, r2 = NAN;
                    if (mc[x]) {
                        const float Rf = (float)acc[Q_R * W + x];  /* boxFilter: input depth */
                        const float Nf = (float)acc[Q_N * W + x];
                        const double Nd = (double)Nf;
                        const double S2 = acc[Q_S2 * W + x], R2s = acc[Q_R2 * W + x]; /* sqrBoxFilter: float64 */
                        if (model == MODEL_GAIN_OFFSET) {
                            /* kernel_model.py:338-351 */
                            const float Sf = (float)acc[Q_S * W + x], Pf = (float)acc[Q_P * W + x];
                            if (may_inpaint) {
                                keepS[(size_t)y * W + x] = Sf, keepR[(size_t)y * W + x] = Rf, keepN[(size_t)y * W + x] = Nf;
                            }
                            const float num = (float)(Nf * Pf) - (float)(Sf * Rf);
                            const double den = (Nd * S2) - (double)(float)(Sf * Sf);
                            g = (float)((double)num / den);
                            o = (float)((float)(Rf - (float)(g * Sf)) / Nf);
                            if (with_r2) { /* kernel_model.py:179,189-195,203,212-213 */
                                const double sstot = (Nd * R2s) - (double)(float)(Rf * Rf);
                                const double A = (double)(float)(g * g) * S2;
                                const float B = (float)((float)(2.f * (float)(g * o)) * Sf);
                                const float C = (float)((float)(2.f * g) * Pf);
                                const float D = (float)((float)(2.f * o) * Rf);
                                const float F = (float)(Nf * (float)(o * o));
                                double ssres = A + (double)B;
                                ssres = ssres - (double)C;
                                ssres = ssres - (double)D;
                                ssres = ssres + R2s;
                                ssres = ssres + (double)F;
                                ssres = ssres * Nd;
                                r2 = 1.f - (float)(ssres / sstot);
                            }
                        } else if (model == MODEL_GAIN_BLK_OFFSET) {
                            /* kernel_model.py:265 with float64 src_sum, then :301-302 */
                            const double Sd = acc[Q_S * W + x], Pd = acc[Q_P * W + x];
                            const float gp = (float)((double)Rf / Sd);
                            if (with_r2) { /* kernel_model.py:179,201,203,212-213 */
                                const double sstot = (Nd * R2s) - (double)(float)(Rf * Rf);
                                double ssres = (double)(float)(gp * gp) * S2;
                                ssres = ssres - ((double)(float)(2.f * gp) * Pd);
                                ssres = ssres + R2s;
                                ssres = ssres * Nd;
                                r2 = 1.f - (float)(ssres / sstot);
                            }
                            o = (float)((double)gp * n1);
                            g = (float)((double)gp * n0);
                        } else {
                            /* kernel_model.py:262-265 */
                            const float Sf = (float)acc[Q_S * W + x], Pf = (float)acc[Q_P * W + x];
                            g = Rf / Sf;
                            o = 0.f;
                            if (with_r2) {
                                const double sstot = (Nd * R2s) - (double)(float)(Rf * Rf);
                                double ssres = (double)(float)(g * g) * S2;
                                ssres = ssres - (double)(float)((float)(2.f * g) * Pf);
                                ssres = ssres + R2s;
                                ssres = ssres * Nd;
                                r2 = 1.f - (float)(ssres / sstot);
                            }
                        }
                        if (model == MODEL_GAIN_OFFSET && has_thresh && !((r2 > thresh) && (g > 0.f))) ++fails; /* :363 */
                    }
                    const size_t idx = (size_t)y * W + x;
                    if (params_out) {
                        params_out[idx] = g;
                        params_out[plane + idx] = o;
                        if (with_r2) params_out[2 * plane + idx] = r2;
                    }
                    if (corr_out) corr_out[idx] = (float)((float)(g * sc[x]) + o); /* kernel_model.py:461 */
                }
            }
        }
        free(ws.ring);
        free(ws.zrow);
        free(mring);
        free(sring);
        free(acc);
    }
    if (!err && may_inpaint && fails > 0) {
        /* kernel_model.py:363-371 */
        unsigned char* r2_mask = (unsigned char*)malloc(plane);
        if (!r2_mask) err = 1;
        else {
            float *pg = params_out, *po = params_out + plane, *pr = params_out + 2 * plane;
            for (size_t i = 0; i < plane; ++i) r2_mask[i] = (pr[i] > thresh) && (pg[i] > 0.f); /* NaN outside mask -> 0 */
            float* filled = (float*)malloc(sizeof(float) * plane);
            if (!filled) err = 1;
            else {
                memcpy(filled, po, sizeof(float) * plane);
                if (hk_oracle_fill_nodata(filled, r2_mask, H, W, 100.0)) err = 1;
                /* param_ra.mask = mask (:367): masked pixels back to NaN; redo = ~r2_mask & mask (:370) */
                for (size_t i = 0; i < plane && !err; ++i) {
                    const int valid = px_valid(src[i], snd_mode, snd) && px_valid(ref[i], rnd_mode, rnd);
                    if (!valid) continue; /* parameters already NaN there */
                    po[i] = filled[i];
                    if (!r2_mask[i]) pg[i] = (float)((float)(keepR[i] - (float)(keepN[i] * po[i])) / keepS[i]); /* :371 */
                    if (corr_out) corr_out[i] = (float)((float)(pg[i] * src[i]) + po[i]);
                }
                free(filled);
            }
            free(r2_mask);
        }
    }
    free(own_params), free(keepS), free(keepR), free(keepN);
    if (fail_count) *fail_count = fails;
    return err ? -1 : 0;
}

/*
 * rasterio.fill.fillnodata(image, mask, max_search_distance, smoothing_iterations=0) == GDALFillNodata
 * (kernel_model.py:366).  GDAL is not in /root/reference: its published algorithm (gdal/alg/rasterfill.cpp) is restated
 * exactly as in oracle_np.fill_nodata -- PARITY WITH GDAL UNPINNED.  image is updated in place.
 */
static void quad_check(double* qd, double* qv, int tx, int64_t ty, int ox, int oy, float tv) {
    if (ty == INT64_MAX) return;
    const double dx = (double)tx - (double)ox, dy = (double)ty - (double)oy;
    const double d2 = dx * dx + dy * dy;
    if (d2 < (*qd) * (*qd)) {
        *qd = sqrt(d2);
        *qv = (double)tv;
    }
}

int hk_oracle_fill_nodata(float* image, const unsigned char* mask, int H, int W, double max_search_distance) {
    const size_t n = (size_t)H * W;
    const int md = (int)floor(max_search_distance);
    int64_t* top_y = (int64_t*)malloc(sizeof(int64_t) * n);
    int64_t* bot_y = (int64_t*)malloc(sizeof(int64_t) * n);
    float* top_v = (float*)malloc(sizeof(float) * n);
    float* bot_v = (float*)malloc(sizeof(float) * n);
    float* out = (float*)malloc(sizeof(float) * n);
    if (!top_y || !bot_y || !top_v || !bot_v || !out) {
        free(top_y), free(bot_y), free(top_v), free(bot_v), free(out);
        return -1;
    }
#pragma omp parallel for
    for (int x = 0; x < W; ++x) {
        int64_t last_y = INT64_MAX;
        float last_v = 0.f;
        for (int y = 0; y < H; ++y) { /* nearest source at or above */
            const size_t i = (size_t)y * W + x;
            if (mask[i]) last_y = y, last_v = image[i];
            else if (last_y != INT64_MAX && y > md + last_y) last_y = INT64_MAX;
            top_y[i] = last_y, top_v[i] = last_v;
        }
        last_y = INT64_MAX, last_v = 0.f;
        for (int y = H - 1; y >= 0; --y) { /* nearest source strictly below */
            const size_t i = (size_t)y * W + x;
            bot_y[i] = last_y, bot_v[i] = last_v;
            if (mask[i]) last_y = y, last_v = image[i];
            else if (last_y != INT64_MAX && last_y - y > md) last_y = INT64_MAX;
        }
    }
#pragma omp parallel for schedule(dynamic, 4)
    for (int y = 0; y < H; ++y) {
        for (int x = 0; x < W; ++x) {
            const size_t i = (size_t)y * W + x;
            out[i] = image[i];
            if (mask[i]) continue;
            double qd[4], qv[4] = {0, 0, 0, 0};
            for (int q = 0; q < 4; ++q) qd[q] = max_search_distance + 1.0;
            int this_max = md;
            for (int step = 0; step <= this_max; ++step) {
                const int lx = x - step > 0 ? x - step : 0, rx = x + step < W - 1 ? x + step : W - 1;
                const size_t il = (size_t)y * W + lx, ir = (size_t)y * W + rx;
                quad_check(&qd[0], &qv[0], lx, top_y[il], x, y, top_v[il]);
                quad_check(&qd[1], &qv[1], lx, bot_y[il], x, y, bot_v[il]);
                if (step == 0) continue;
                quad_check(&qd[2], &qv[2], rx, top_y[ir], x, y, top_v[ir]);
                quad_check(&qd[3], &qv[3], rx, bot_y[ir], x, y, bot_v[ir]);
                if ((step & 3) == 0) this_max = (int)floor(fmax(fmax(qd[0], qd[1]), fmax(qd[2], qd[3])));
            }
            double wsum = 0.0, vsum = 0.0;
            int has = 0;
            for (int q = 0; q < 4; ++q) {
                if (qd[q] <= max_search_distance) {
                    const double wgt = 1.0 / qd[q];
                    has = wgt != 0.0;
                    wsum += wgt;
                    vsum += qv[q] * wgt;
                }
            }
            if (has) out[i] = (float)(vsum / wsum);
        }
    }
    memcpy(image, out, sizeof(float) * n);
    free(top_y), free(bot_y), free(top_v), free(bot_v), free(out);
    return 0;
}

/* KernelModel.apply alone (kernel_model.py:461) */
void hk_oracle_apply(const float* src, const float* params, int H, int W, float* out) {
    const size_t plane = (size_t)H * W;
#pragma omp parallel for
    for (int64_t i = 0; i < (int64_t)plane; ++i) out[i] = (float)((float)(params[i] * src[i]) + params[plane + i]);
}

/* quickselect with 3-way partitioning: k-th smallest of a[0..n) (a is permuted) */
static float select_kth(float* a, size_t n, size_t k) {
    size_t lo = 0, hi = n; /* candidates in [lo, hi) */
    while (hi - lo > 1) {
        const float x = a[lo], y = a[lo + (hi - lo) / 2], z = a[hi - 1];
        const float p = x < y ? (y < z ? y : (x < z ? z : x)) : (x < z ? x : (y < z ? z : y)); /* median of 3 */
        size_t lt = lo, i = lo, gt = hi;
        while (i < gt) {
            if (a[i] < p) {
                const float t = a[lt];
                a[lt++] = a[i];
                a[i++] = t;
            } else if (a[i] > p) {
                const float t = a[--gt];
                a[gt] = a[i];
                a[i] = t;
            } else {
                ++i;
            }
        }
        if (k < lt) hi = lt;
        else if (k >= gt) lo = gt;
        else return p;
    }
    return a[lo];
}

static double percentile1(float* a, size_t n) {
    /* np.percentile(a, 1), method 'linear': virtual index 0.01 (n-1), numpy's _lerp */
    const double v = 0.01 * (double)(n - 1);
    const size_t k0 = (size_t)floor(v), k1 = k0 + 1 < n ? k0 + 1 : n - 1;
    const double t = v - (double)k0;
    const double lo = (double)select_kth(a, n, k0);
    const double hi = (double)select_kth(a, n, k1);
    const double d = hi - lo;
    return t >= 0.5 ? hi - d * (1.0 - t) : lo + d * t;
}

/*
 * KernelModel._fit_block_norm (kernel_model.py:216-229), FLOAT64 flavour: exact two-pass population std and the
 * exact order statistics -- what the GPU computes.  numpy itself runs these in float32 pairwise arithmetic; the two
 * agree to ~5e-7 relative (the numpy-faithful values are in oracle_np.fit_block_norm / the goldens).
 */
int hk_oracle_block_norm(const float* src, int snd_mode, float snd, const float* ref, int rnd_mode, float rnd, int H,
                         int W, double norm_out[2]) {
    const size_t total = (size_t)H * W;
    float* s = (float*)malloc(sizeof(float) * total);
    float* r = (float*)malloc(sizeof(float) * total);
    if (!s || !r) {
        free(s);
        free(r);
        return -1;
    }
    size_t n = 0;
    double sum_s = 0.0, sum_r = 0.0;
    for (size_t i = 0; i < total; ++i) {
        if (px_valid(src[i], snd_mode, snd) && px_valid(ref[i], rnd_mode, rnd)) {
            s[n] = src[i];
            r[n] = ref[i];
            sum_s += (double)src[i];
            sum_r += (double)ref[i];
            ++n;
        }
    }
    norm_out[0] = norm_out[1] = 0.0;
    if (n > 0) {
        const double mean_s = sum_s / (double)n, mean_r = sum_r / (double)n;
        double vs = 0.0, vr = 0.0;
        for (size_t i = 0; i < n; ++i) {
            const double ds = (double)s[i] - mean_s, dr = (double)r[i] - mean_r;
            vs += ds * ds;
            vr += dr * dr;
        }
        const double n0 = sqrt(vr / (double)n) / sqrt(vs / (double)n);
        const double pr = percentile1(r, n), ps = percentile1(s, n);
        norm_out[0] = n0;
        norm_out[1] = pr - ps * n0;
    }
    free(s);
    free(r);
    return 0;
}

int hk_oracle_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
