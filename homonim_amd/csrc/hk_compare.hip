// hk_compare.hip -- the masked sums behind the accuracy statistics of homonim/compare.py:243-255 (RasterCompare.process,
// get_block_sums): per band, over the pixels valid in BOTH rasters,
//     [ sum s, sum r, sum f32(s*s), sum f32(r*r), sum f32(s*r), sum f32(f32(r-s)^2), count ].
// The reference forms every per-pixel term in float32 (numpy on float32 arrays) and adds them up with numpy's float32
// pairwise `.sum()`; here the same float32 terms are accumulated in float64 (fixed order: deterministic), which is the
// exact sum to ~1e-16 -- the reference's own result differs from it by its float32 summation error (~1e-7 relative).
//
// HBM-bound reduction: 8 bytes read per pixel*band, nothing written.  One pass; grid = COMPARE_BLOCKS x bands, each
// workgroup walks whole rows with 16-byte loads, wave butterfly + LDS across the four waves, one partial per workgroup;
// a second one-wave kernel per band adds the partials in index order.
#include <hip/hip_runtime.h>

#include "hk_kernels.h"

namespace hk {

namespace {

constexpr int COMPARE_THREADS = 256;
constexpr int COMPARE_BLOCKS = 2048;  // workgroups per band (8 per CU)
constexpr int NS = 7;

__device__ __forceinline__ bool cvalid(float v, int mode, float nodata) {
    return mode == 0 ? true : (mode == 1 ? !(v != v) : !(v == nodata));
}

__device__ __forceinline__ double wave_sum_f64(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

struct Acc {
    double s = 0.0, r = 0.0, s2 = 0.0, r2 = 0.0, sr = 0.0, d2 = 0.0;
    unsigned n = 0;
    __device__ __forceinline__ void add(float sv, float rv, bool m) {
        // compare.py:246-247: src_array[~mask] = 0, ref_array[~mask] = 0 -- masked pixels add zeros
        const float a = m ? sv : 0.f, b = m ? rv : 0.f;
        const float d = __fsub_rn(b, a);
        s += (double)a, r += (double)b;
        s2 += (double)__fmul_rn(a, a), r2 += (double)__fmul_rn(b, b), sr += (double)__fmul_rn(a, b);
        d2 += (double)__fmul_rn(d, d);
        n += m ? 1u : 0u;
    }
};

}  // namespace

size_t compare_workspace_bytes(int n_bands) { return (size_t)n_bands * COMPARE_BLOCKS * NS * sizeof(double); }

__global__ void __launch_bounds__(COMPARE_THREADS) compare_partial_kernel(const CompareArgs a, double* __restrict__ partials) {
    const int band = blockIdx.y;
    const float* __restrict__ sp = a.src + (long long)band * a.src_band_stride;
    const float* __restrict__ rp = a.ref + (long long)band * a.ref_band_stride;
    Acc acc;
    unsigned long long n_total = 0;
    const int wq = a.width / 4;  // whole 4-pixel groups; rows are 16-byte aligned when vec_ok
    for (int y = blockIdx.x; y < a.height; y += gridDim.x) {
        const float* __restrict__ srow = sp + (long long)y * a.src_stride;
        const float* __restrict__ rrow = rp + (long long)y * a.ref_stride;
        int x0 = 0;
        if (a.vec_ok) {
            const float4* __restrict__ s4 = reinterpret_cast<const float4*>(srow);
            const float4* __restrict__ r4 = reinterpret_cast<const float4*>(rrow);
            for (int q = threadIdx.x; q < wq; q += COMPARE_THREADS) {
                const float4 s = s4[q], r = r4[q];
                const float sv[4] = {s.x, s.y, s.z, s.w}, rv[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    acc.add(sv[i], rv[i], cvalid(sv[i], a.src_nd_mode, a.src_nodata) && cvalid(rv[i], a.ref_nd_mode, a.ref_nodata));
            }
            x0 = wq * 4;
        }
        for (int x = x0 + threadIdx.x; x < a.width; x += COMPARE_THREADS) {
            const float s = srow[x], r = rrow[x];
            acc.add(s, r, cvalid(s, a.src_nd_mode, a.src_nodata) && cvalid(r, a.ref_nd_mode, a.ref_nodata));
        }
        n_total += acc.n, acc.n = 0;  // a row of one thread holds < 2^32 pixels
    }
    double v[NS] = {acc.s, acc.r, acc.s2, acc.r2, acc.sr, acc.d2, (double)n_total};  // counts < 2^53: exact
    __shared__ double red[COMPARE_THREADS / 64][NS];
#pragma unroll
    for (int k = 0; k < NS; ++k) v[k] = wave_sum_f64(v[k]);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0)
        for (int k = 0; k < NS; ++k) red[wave][k] = v[k];
    __syncthreads();
    if (threadIdx.x < NS) {
        double t = 0.0;
        for (int w = 0; w < COMPARE_THREADS / 64; ++w) t += red[w][threadIdx.x];
        partials[((size_t)band * gridDim.x + blockIdx.x) * NS + threadIdx.x] = t;
    }
}

__global__ void __launch_bounds__(64) compare_final_kernel(const double* __restrict__ partials, int n_partials,
                                                            double* __restrict__ sums_out) {
    const int band = blockIdx.x;
    const double* __restrict__ p = partials + (size_t)band * n_partials * NS;
    double v[NS] = {0, 0, 0, 0, 0, 0, 0};
    for (int i = threadIdx.x; i < n_partials; i += 64)
        for (int k = 0; k < NS; ++k) v[k] += p[(size_t)i * NS + k];
    for (int k = 0; k < NS; ++k) v[k] = wave_sum_f64(v[k]);
    if (threadIdx.x == 0)
        for (int k = 0; k < NS; ++k) sums_out[band * NS + k] = v[k];
}

hipError_t launch_compare_sums(const CompareArgs& a_in, void* workspace, double* sums_out, hipStream_t stream) {
    CompareArgs a = a_in;
    a.vec_ok = ((a.src_stride | a.ref_stride | a.src_band_stride | a.ref_band_stride) % 4 == 0) &&
               ((((uintptr_t)a.src) | ((uintptr_t)a.ref)) % 16 == 0);
    const int blocks = a.height < COMPARE_BLOCKS ? a.height : COMPARE_BLOCKS;
    double* partials = static_cast<double*>(workspace);
    HK_LAUNCH(compare_partial_kernel, dim3(blocks, a.n_bands), dim3(COMPARE_THREADS), 0, stream, a, partials);
    HK_LAUNCH(compare_final_kernel, dim3(a.n_bands), dim3(64), 0, stream, partials, blocks, sums_out);
    return hipGetLastError();
}

}  // namespace hk
