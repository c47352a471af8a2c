// hk_norm.hip -- block normalisation statistics of the gain-blk-offset model on gfx950.
//
// Reference: KernelModel._fit_block_norm (homonim/kernel_model.py:216-229)
//     mask = ref.mask & src.mask
//     norm[0] = np.std(ref[mask]) / np.std(src[mask])
//     norm[1] = np.percentile(ref[mask], 1) - np.percentile(src[mask], 1) * norm[0]
// np.std = population std; np.percentile(., 1) = linear interpolation between the order statistics at
// floor(0.01 (n-1)) and the next one.  Here: exact float64 two-pass mean / variance (deterministic block partials,
// fixed reduction order) and an EXACT order-statistic select: 3-level radix histograms (11 + 11 + 10 bits of the
// order-preserving uint32 image of the float) -- no sort, no candidate buffer, 3 streaming passes, HBM-bound.
// numpy runs the same statistics in float32 pairwise arithmetic; the two agree to ~5e-7 relative.
#include "hk_kernels.h"

namespace hk {

constexpr int NORM_BLOCKS = 512;  // partial-reduction blocks per band
constexpr int NORM_THREADS = 256;
constexpr int L1_BITS = 11, L2_BITS = 11, L3_BITS = 10;
constexpr int L1_BINS = 1 << L1_BITS, L2_BINS = 1 << L2_BITS, L3_BINS = 1 << L3_BITS;

struct Sel {
    unsigned prefix;           // key bits fixed so far (right-aligned)
    unsigned long long rank;   // rank still to resolve inside that prefix
};

struct NormWS {  // one per band
    unsigned long long pn[NORM_BLOCKS];
    double ps[NORM_BLOCKS], pr[NORM_BLOCKS];
    double pvs[NORM_BLOCKS], pvr[NORM_BLOCKS];
    unsigned hist1[2][L1_BINS];       // [src|ref]
    unsigned hist2[2][2][L2_BINS];    // [src|ref][rank k0|k1]
    unsigned hist3[2][2][L3_BINS];
    unsigned long long n;
    double mean_s, mean_r, var_s, var_r, frac;
    Sel sel[2][2];
    float val[2][2];
};

size_t norm_workspace_bytes(int n_bands) { return sizeof(NormWS) * (size_t)n_bands; }

__device__ __forceinline__ unsigned f2key(float f) {  // order-preserving float -> uint32
    const unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float key2f(unsigned k) {
    const unsigned u = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
    return __uint_as_float(u);
}
__device__ __forceinline__ bool nvalid(float v, int mode, float nodata) {
    return mode == 0 ? true : (mode == 1 ? !(v != v) : !(v == nodata));
}

// Deterministic block reduction (fixed tree) of a double / u64 through LDS.
template <typename T>
__device__ __forceinline__ T block_reduce(T v, T* sh) {
    const int t = threadIdx.x;
    sh[t] = v;
    __syncthreads();
    for (int d = NORM_THREADS / 2; d > 0; d >>= 1) {
        if (t < d) sh[t] = sh[t] + sh[t + d];
        __syncthreads();
    }
    const T r = sh[0];
    __syncthreads();
    return r;
}

// PASS: 0 = count/sum + level-1 histograms; 1 = squared deviations + level-2; 2 = level-3.
template <int PASS>
__global__ void __launch_bounds__(NORM_THREADS) norm_pass_kernel(const NormArgs a, NormWS* __restrict__ ws_all) {
    constexpr int NH = PASS == 0 ? 2 * L1_BINS : (PASS == 1 ? 4 * L2_BINS : 4 * L3_BINS);
    __shared__ unsigned hist[NH];
    __shared__ double shd[NORM_THREADS];
    __shared__ unsigned long long shn[NORM_THREADS];
    const int band = blockIdx.y;
    NormWS& ws = ws_all[band];
    if (PASS > 0 && ws.n == 0) return;
    for (int i = threadIdx.x; i < NH; i += NORM_THREADS) hist[i] = 0;
    __syncthreads();

    const float* __restrict__ sp = a.src + (long long)band * a.band_stride;
    const float* __restrict__ rp = a.ref + (long long)band * a.band_stride;
    double mean_s = 0.0, mean_r = 0.0;
    unsigned pfx[2][2] = {{0, 0}, {0, 0}};
    if (PASS >= 1) {
        mean_s = ws.mean_s;
        mean_r = ws.mean_r;
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int k = 0; k < 2; ++k) pfx[q][k] = ws.sel[q][k].prefix;
    }

    unsigned long long n = 0;
    double acc_s = 0.0, acc_r = 0.0;
    const int wq = (a.width + PX - 1) / PX;
    for (int y = blockIdx.x; y < a.height; y += gridDim.x) {
        const long long row = (long long)y * a.stride;
        for (int xq = threadIdx.x; xq < wq; xq += NORM_THREADS) {
            const int x = xq * PX;
            float s[PX], r[PX];
            if (x + PX <= a.width) {
                const float4 s4 = *reinterpret_cast<const float4*>(sp + row + x);
                const float4 r4 = *reinterpret_cast<const float4*>(rp + row + x);
                s[0] = s4.x, s[1] = s4.y, s[2] = s4.z, s[3] = s4.w;
                r[0] = r4.x, r[1] = r4.y, r[2] = r4.z, r[3] = r4.w;
            } else {
#pragma unroll
                for (int i = 0; i < PX; ++i) {
                    const bool in = x + i < a.width;
                    s[i] = in ? sp[row + x + i] : 0.f;
                    r[i] = in ? rp[row + x + i] : 0.f;
                }
            }
#pragma unroll
            for (int i = 0; i < PX; ++i) {
                const bool m = x + i < a.width && nvalid(s[i], a.src_nd_mode, a.src_nodata) &&
                               nvalid(r[i], a.ref_nd_mode, a.ref_nodata);
                if (!m) continue;
                const unsigned ks = f2key(s[i]), kr = f2key(r[i]);
                if (PASS == 0) {
                    ++n;
                    acc_s += (double)s[i];
                    acc_r += (double)r[i];
                    atomicAdd(&hist[ks >> (32 - L1_BITS)], 1u);
                    atomicAdd(&hist[L1_BINS + (kr >> (32 - L1_BITS))], 1u);
                } else if (PASS == 1) {
                    const double ds = (double)s[i] - mean_s, dr = (double)r[i] - mean_r;
                    acc_s += ds * ds;
                    acc_r += dr * dr;
#pragma unroll
                    for (int k = 0; k < 2; ++k) {
                        if ((ks >> (32 - L1_BITS)) == pfx[0][k])
                            atomicAdd(&hist[(0 * 2 + k) * L2_BINS + ((ks >> L3_BITS) & (L2_BINS - 1))], 1u);
                        if ((kr >> (32 - L1_BITS)) == pfx[1][k])
                            atomicAdd(&hist[(1 * 2 + k) * L2_BINS + ((kr >> L3_BITS) & (L2_BINS - 1))], 1u);
                    }
                } else {
#pragma unroll
                    for (int k = 0; k < 2; ++k) {
                        if ((ks >> L3_BITS) == pfx[0][k]) atomicAdd(&hist[(0 * 2 + k) * L3_BINS + (ks & (L3_BINS - 1))], 1u);
                        if ((kr >> L3_BITS) == pfx[1][k]) atomicAdd(&hist[(1 * 2 + k) * L3_BINS + (kr & (L3_BINS - 1))], 1u);
                    }
                }
            }
        }
    }
    __syncthreads();
    unsigned* gh = PASS == 0 ? &ws.hist1[0][0] : (PASS == 1 ? &ws.hist2[0][0][0] : &ws.hist3[0][0][0]);
    for (int i = threadIdx.x; i < NH; i += NORM_THREADS)
        if (hist[i]) atomicAdd(gh + i, hist[i]);  // integer atomics: order-independent result
    if (PASS <= 1) {
        const double rs = block_reduce<double>(acc_s, shd);
        const double rr = block_reduce<double>(acc_r, shd);
        if (PASS == 0) {
            const unsigned long long rn = block_reduce<unsigned long long>(n, shn);
            if (threadIdx.x == 0) ws.pn[blockIdx.x] = rn, ws.ps[blockIdx.x] = rs, ws.pr[blockIdx.x] = rr;
        } else if (threadIdx.x == 0) {
            ws.pvs[blockIdx.x] = rs, ws.pvr[blockIdx.x] = rr;
        }
    }
}

// Find the bin holding rank `rank` in hist[0..nbins): returns bin, and the rank inside it.
__device__ void select_bin(const unsigned* __restrict__ h, int nbins, unsigned long long rank, unsigned* bin_out,
                           unsigned long long* rank_out, unsigned long long* sh /* NORM_THREADS */) {
    const int t = threadIdx.x;
    const int per = nbins / NORM_THREADS;
    unsigned long long local = 0;
    for (int i = 0; i < per; ++i) local += h[t * per + i];
    sh[t] = local;
    __syncthreads();
    if (t == 0) {
        unsigned long long cum = 0;
        int tt = 0;
        for (; tt < NORM_THREADS - 1; ++tt) {
            if (cum + sh[tt] > rank) break;
            cum += sh[tt];
        }
        int b = tt * per;
        for (int i = 0; i < per - 1; ++i, ++b) {
            if (cum + h[b] > rank) break;
            cum += h[b];
        }
        *bin_out = (unsigned)b;
        *rank_out = rank - cum;
    }
    __syncthreads();
}

// STAGE 0: after pass 0 (n, means, ranks, level-1 select); 1: after pass 1 (variances, level-2 select);
// 2: after pass 2 (level-3 select -> order statistics -> norm).
template <int STAGE>
__global__ void __launch_bounds__(NORM_THREADS) norm_finalize_kernel(NormWS* __restrict__ ws_all, double* __restrict__ norm_out) {
    __shared__ unsigned long long sh[NORM_THREADS];
    __shared__ unsigned bin;
    __shared__ unsigned long long rk;
    NormWS& ws = ws_all[blockIdx.x];
    if (STAGE == 0) {
        if (threadIdx.x == 0) {
            unsigned long long n = 0;
            double ss = 0.0, sr = 0.0;
            for (int i = 0; i < NORM_BLOCKS; ++i) n += ws.pn[i], ss += ws.ps[i], sr += ws.pr[i];
            ws.n = n;
            if (n > 0) {
                ws.mean_s = ss / (double)n;
                ws.mean_r = sr / (double)n;
                const double v = 0.01 * (double)(n - 1);
                const unsigned long long k0 = (unsigned long long)floor(v);
                ws.frac = v - (double)k0;
                ws.sel[0][0].rank = ws.sel[1][0].rank = k0;
                ws.sel[0][1].rank = ws.sel[1][1].rank = (k0 + 1 < n) ? k0 + 1 : n - 1;
            } else {
                norm_out[2 * blockIdx.x] = 0.0;  // kernel_model.py:223-226
                norm_out[2 * blockIdx.x + 1] = 0.0;
            }
        }
        __syncthreads();
        if (ws.n == 0) return;
    } else if (ws.n == 0) {
        return;
    }
    if (STAGE == 1 && threadIdx.x == 0) {
        double vs = 0.0, vr = 0.0;
        for (int i = 0; i < NORM_BLOCKS; ++i) vs += ws.pvs[i], vr += ws.pvr[i];
        ws.var_s = vs / (double)ws.n;
        ws.var_r = vr / (double)ws.n;
    }
    for (int q = 0; q < 2; ++q) {
        for (int k = 0; k < 2; ++k) {
            const unsigned* h = STAGE == 0 ? ws.hist1[q] : (STAGE == 1 ? ws.hist2[q][k] : ws.hist3[q][k]);
            const int nb = STAGE == 0 ? L1_BINS : (STAGE == 1 ? L2_BINS : L3_BINS);
            select_bin(h, nb, ws.sel[q][k].rank, &bin, &rk, sh);
            if (threadIdx.x == 0) {
                const int bits = STAGE == 0 ? 0 : (STAGE == 1 ? L2_BITS : L3_BITS);
                ws.sel[q][k].prefix = (STAGE == 0) ? bin : ((ws.sel[q][k].prefix << bits) | bin);
                ws.sel[q][k].rank = rk;
                if (STAGE == 2) ws.val[q][k] = key2f(ws.sel[q][k].prefix);
            }
            __syncthreads();
        }
    }
    if (STAGE == 2 && threadIdx.x == 0) {
        const double n0 = sqrt(ws.var_r) / sqrt(ws.var_s);
        double pct[2];
        for (int q = 0; q < 2; ++q) {
            // numpy _lerp (lib/_function_base_impl.py): a + (b-a)*t, and b - (b-a)*(1-t) where t >= 0.5
            const double lo = (double)ws.val[q][0], hi = (double)ws.val[q][1], t = ws.frac;
            const double d = hi - lo;
            pct[q] = t >= 0.5 ? hi - d * (1.0 - t) : lo + d * t;
        }
        norm_out[2 * blockIdx.x] = n0;
        norm_out[2 * blockIdx.x + 1] = pct[1] - pct[0] * n0;
    }
}

hipError_t launch_block_norm(const NormArgs& a, void* workspace, double* norm_out, hipStream_t stream) {
    NormWS* ws = reinterpret_cast<NormWS*>(workspace);
    hipError_t e = hipMemsetAsync(ws, 0, sizeof(NormWS) * (size_t)a.n_bands, stream);
    if (e != hipSuccess) return e;
    const dim3 grid(NORM_BLOCKS, a.n_bands), block(NORM_THREADS);
    hipLaunchKernelGGL(norm_pass_kernel<0>, grid, block, 0, stream, a, ws);
    hipLaunchKernelGGL(norm_finalize_kernel<0>, dim3(a.n_bands), block, 0, stream, ws, norm_out);
    hipLaunchKernelGGL(norm_pass_kernel<1>, grid, block, 0, stream, a, ws);
    hipLaunchKernelGGL(norm_finalize_kernel<1>, dim3(a.n_bands), block, 0, stream, ws, norm_out);
    hipLaunchKernelGGL(norm_pass_kernel<2>, grid, block, 0, stream, a, ws);
    hipLaunchKernelGGL(norm_finalize_kernel<2>, dim3(a.n_bands), block, 0, stream, ws, norm_out);
    return hipGetLastError();
}

}  // namespace hk
