// hk_norm.hip -- block normalisation statistics of the gain-blk-offset model on gfx950.
//
// Reference: KernelModel._fit_block_norm (homonim/kernel_model.py:216-229)
//     mask = ref.mask & src.mask
//     norm[0] = np.std(ref[mask]) / np.std(src[mask])
//     norm[1] = np.percentile(ref[mask], 1) - np.percentile(src[mask], 1) * norm[0]
// np.std = population std; np.percentile(., 1) = linear interpolation between the order statistics at
// floor(0.01 (n-1)) and the next one.
//
// Here the moments are float64 (deterministic per-wave partials reduced in a fixed order) and the order statistics
// are EXACT.  numpy runs the same statistics in float32 pairwise arithmetic; the two agree to ~5e-7 relative.
//
// ONE streaming pass over src+ref (8 B per pixel, HBM-bound) instead of a sort:
//   1. sample:  a 4096-pixel stratified sample per band, taken by one small workgroup: its mean is the shift of the moment
//               sums and, per raster, two of its order statistics (an LDS radix select) are the pivots that bracket the
//               1st percentile (+-4 sigma of the sample quantile);
//   2. pass:    every wave streams its 1 KB chunks: count, shifted first/second moments, number of values below the low
//               pivot, and the ~1 % of values between the pivots compacted through lane-private LDS queues into a
//               small HBM buffer (one global atomic per flush of the wave's queues);
//   3. select:  the two ranks are resolved inside the compacted buffer by a 3-level radix select (11+11+10 bits of the
//               order-preserving uint32 image of the float, taken relative to the pivot window) -- integer histograms only,
//               so the result is deterministic.
// If the pivots miss (or the buffer overflows) a device-side flag routes the band through the same radix select over
// the full rasters (three more passes); those kernels are always launched and exit immediately otherwise.
// Every kernel here is sized to be PLACED beside the fit kernels of other streams (<= 24 KB of LDS per workgroup): a block's
// statistics are a chain of dependent launches, and a workgroup that has to wait for a CU to drain stalls the whole chain.
// launch_block_norm_split(): the same statistics for a block whose rows are spread over several ranks (see there).
#include "hk_kernels.h"

#include <type_traits>

namespace hk {

constexpr int NORM_THREADS = 256;
constexpr int L1_BITS = 11, L2_BITS = 11, L3_BITS = 10;
constexpr int L1_BINS = 1 << L1_BITS, L2_BINS = 1 << L2_BITS, L3_BINS = 1 << L3_BITS;
constexpr int SAMPLE_N = 4096;     // sample size per band
#ifndef HK_NORM_LDS_PAD_DEFAULT
#define HK_NORM_LDS_PAD_DEFAULT 0
#endif
#ifndef HK_NORM_ABLATE
#define HK_NORM_ABLATE 0  // timing experiments only (wrong results): 1 no moments, 2 no compaction queues
#endif
#ifndef HK_PASS_WAVES
#define HK_PASS_WAVES 2048
#endif
constexpr int PASS_WAVES = HK_PASS_WAVES;  // most waves per band in the streaming pass (= the number of partial sums kept)
constexpr int FB_BLOCKS = 512;     // workgroups per band of the fallback passes
constexpr int MID_BLOCKS = 256;    // workgroups per compacted buffer of the regular select passes

struct Sel {
    unsigned prefix;          // key bits fixed so far (right-aligned)
    unsigned long long rank;  // rank still to resolve inside that prefix
};

struct NormWS {  // one per band
    // sample stage
    float lo[2], hi[2];  // [src|ref] pivots: values in [lo, hi] are compacted
    double shift[2];
    // streaming pass
    unsigned long long pn[PASS_WAVES], pbelow[2][PASS_WAVES];
    double p1[2][PASS_WAVES], p2[2][PASS_WAVES];
    unsigned mid_count[2];
    // statistics
    unsigned long long n;
    double mean[2], var[2], frac;
    unsigned long long k[2];  // the two ranks (k0, k0+1 clipped)
    int fallback, done;
    // radix select state (shared by the compacted-buffer select and the fallback)
    Sel sel[2][2];  // [src|ref][rank k0|k1]
    // the select's keys: (f2key(v) - kbase) << ksh.  Over the compacted buffers the window [lo, hi] is stretched over all 32
    // bits, so the first digit spreads over the whole histogram instead of one hot bin; 0 / 0 = plain keys (full rasters)
    unsigned kbase[2];
    int ksh[2];
    float val[2][2];
    unsigned hist1[2][2][L1_BINS], hist2[2][2][L2_BINS], hist3[2][2][L3_BINS];
};

static size_t mid_capacity(long long n_px) { return (size_t)(n_px / 25 + 8192); }  // 4 % of the block + slack
static size_t align256(size_t b) { return (b + 255) / 256 * 256; }

size_t norm_workspace_bytes(int n_bands, int height, int width) {
    return align256(sizeof(NormWS) * (size_t)n_bands) +
           (size_t)n_bands * 2 * align256(mid_capacity((long long)height * width) * sizeof(float));
}

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

// geometry of plane p of the launch: the job's band p, or entry p of a batched launch's plane table (uniform over the workgroup)
struct PlaneRef {
    const float* sp;
    const float* rp;
    long long stride;
    int height, width;
};
__device__ __forceinline__ PlaneRef plane_of(const NormArgs& a, int p) {
    PlaneRef r;
    if (a.planes != nullptr) {
        const __attribute__((address_space(1))) NormPlane* const e = (const __attribute__((address_space(1))) NormPlane*)a.planes + p;  // (the table lives in device memory)
        r.sp = e->src, r.rp = e->ref, r.stride = e->stride, r.height = e->height, r.width = e->width;
    } else {
        r.sp = a.src + (long long)p * a.band_stride, r.rp = a.ref + (long long)p * a.band_stride;
        r.stride = a.stride, r.height = a.height, r.width = a.width;
    }
    return r;
}

// element i of a plane: an explicitly GLOBAL load -- a plane pointer out of a batched launch's table carries no address space the
// compiler could infer and would be read with flat_ instructions (tests/test_isa_cpu.py keeps the library free of them)
__device__ __forceinline__ float plane_at(const float* plane, long long i) {
    return *((__attribute__((address_space(1))) const float*)plane + i);
}

__device__ __forceinline__ double wave_sum(double v) {  // fixed butterfly order -> deterministic
#pragma unroll
    for (int d = WAVE / 2; d > 0; d >>= 1) v += __shfl_xor(v, d);
    return v;
}
__device__ __forceinline__ unsigned long long wave_sum(unsigned long long v) {
#pragma unroll
    for (int d = WAVE / 2; d > 0; d >>= 1) v += __shfl_xor(v, d);
    return v;
}

// Find the bin holding rank `rank` in hist[0..NBINS) -- the first bin whose running count exceeds it (the last bin if
// none does) -- and the rank inside that bin.  One wave: each lane loads a contiguous chunk into registers (independent
// loads) and sums it, an inclusive scan over the lanes finds the chunk, its lane walks its registers.  The lane that found
// it returns true.
template <int NBINS, typename T = unsigned>
__device__ __forceinline__ bool select_bin(const T* __restrict__ h, unsigned long long rank, unsigned* bin_out,
                                           unsigned long long* rank_out) {
    const int lane = threadIdx.x & (WAVE - 1);
    constexpr int PER = NBINS / WAVE;
    unsigned v[PER];
    unsigned long long local = 0;
#pragma unroll
    for (int i = 0; i < PER; ++i) v[i] = h[lane * PER + i];
#pragma unroll
    for (int i = 0; i < PER; ++i) local += v[i];
    unsigned long long incl = local;
#pragma unroll
    for (int d = 1; d < WAVE; d <<= 1) {
        const unsigned long long u = __shfl_up(incl, d, WAVE);
        if (lane >= d) incl += u;
    }
    const unsigned long long hit = __ballot(incl > rank);
    const int owner = hit ? __ffsll(hit) - 1 : WAVE - 1;
    if (lane != owner) return false;
    unsigned long long cum = incl - local;
    int b = 0;
    bool found = false;
#pragma unroll
    for (int i = 0; i < PER - 1; ++i) {
        if (!found) {
            if (cum + v[i] > rank) found = true;
            else cum += v[i], b = i + 1;
        }
    }
    *bin_out = (unsigned)(lane * PER + b);
    *rank_out = rank - cum;
    return true;
}

// ---------------------------------------------------------------------------------------------------------------------
// 1. sample
// Small on purpose -- 256 threads, 24 KB of LDS: on a GPU filled by other streams' fit kernels (16 waves and 128-160 KB of LDS
// per CU) a 1024-thread / 48 KB workgroup waited for a CU to drain, 0.3-0.4 ms per block with four streams in flight.  One
// raster at a time (the sample positions are read twice), 16-bit histogram counters (a sample has 4096 entries).
constexpr int SAMPLE_THREADS = 256;
__global__ void __launch_bounds__(SAMPLE_THREADS) norm_sample_kernel(const NormArgs a, NormWS* __restrict__ ws_all) {
    static_assert(L1_BINS == L2_BINS && L3_BINS <= L1_BINS, "the sample select shares one histogram shape");
    static_assert(SAMPLE_N < 65536, "16-bit histogram counters");
    __shared__ float samp[SAMPLE_N];
    __shared__ unsigned hist[2][L1_BINS / 2];  // one per rank, two 16-bit counters per word
    __shared__ unsigned cnt, spfx[2], srank[2];
    __shared__ int need[2];
    __shared__ double red[SAMPLE_THREADS / WAVE];
    const int band = blockIdx.x, t = threadIdx.x;
    NormWS& ws = ws_all[band];
    const PlaneRef pl = plane_of(a, band);
    const float* __restrict__ sp = pl.sp;
    const float* __restrict__ rp = pl.rp;
    const long long total = (long long)pl.height * pl.width;
    const long long step = total / SAMPLE_N > 0 ? total / SAMPLE_N : 1;
    for (int q = 0; q < 2; ++q) {
        if (t == 0) cnt = 0;
        for (int i = t; i < SAMPLE_N; i += SAMPLE_THREADS) samp[i] = __int_as_float(0x7f800000);  // +inf padding
        __syncthreads();
        double sum = 0.0;
        for (int j = t; j < SAMPLE_N; j += SAMPLE_THREADS) {
            // stratified sample: one pseudo-random pixel per stratum of `step` pixels (a plain stride aliases with the row
            // length -- e.g. every sample in one column of a nodata frame)
            unsigned long long hsh = ((unsigned long long)j + 0x9e3779b97f4a7c15ull) * 0xbf58476d1ce4e5b9ull;
            hsh = (hsh ^ (hsh >> 29)) * 0x94d049bb133111ebull;
            hsh ^= hsh >> 32;
            const long long p = (long long)j * step + (long long)(hsh % (unsigned long long)step);
            if (p < total) {
                const int y = (int)(p / pl.width), x = (int)(p % pl.width);
                const float s = plane_at(sp, (long long)y * pl.stride + x), r = plane_at(rp, (long long)y * pl.stride + x);
                if (nvalid(s, a.src_nd_mode, a.src_nodata) && nvalid(r, a.ref_nd_mode, a.ref_nodata)) {
                    const float v = q ? r : s;
                    samp[atomicAdd(&cnt, 1u)] = v;
                    sum += (double)v;
                }
            }
        }
        __syncthreads();
        const unsigned m = cnt;
        // shift of the moment sums = sample mean (any value works; a close one kills the cancellation): butterfly per
        // wave, thread 0 adds the wave sums in order
        {
            const double wsum = wave_sum(sum);
            if ((t & (WAVE - 1)) == 0) red[t / WAVE] = wsum;
        }
        // sample ranks bracketing the 1st percentile by +-4 sigma of the binomial sample quantile (+ margin); the two order
        // statistics are found by a 3-level radix select in LDS (one histogram and one wave per rank) -- the values a sort
        // would put at those ranks, at a fraction of a bitonic sort's time
        if (t == 0) {
            const double c = 0.01 * (double)m, sd = sqrt(c > 0.0 ? c : 0.0);
            const long long ia = (long long)floor(c - 4.0 * sd) - 2, ib = (long long)ceil(c + 4.0 * sd) + 3;
            need[0] = !(m == 0 || ia <= 0), need[1] = !(m == 0 || ib >= (long long)m - 1);
            srank[0] = need[0] ? (unsigned)ia : 0u, srank[1] = need[1] ? (unsigned)ib : 0u;
            spfx[0] = spfx[1] = 0u;
        }
        __syncthreads();
        for (int level = 0; level < 3; ++level) {
            for (int i = t; i < L1_BINS; i += SAMPLE_THREADS) (&hist[0][0])[i] = 0;
            __syncthreads();
            for (int i = t; i < SAMPLE_N; i += SAMPLE_THREADS) {
                const unsigned key = f2key(samp[i]);
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    unsigned bin;
                    bool in;
                    if (level == 0) bin = key >> (32 - L1_BITS), in = true;
                    else if (level == 1) bin = (key >> L3_BITS) & (L2_BINS - 1), in = (key >> (32 - L1_BITS)) == spfx[k];
                    else bin = key & (L3_BINS - 1), in = (key >> L3_BITS) == spfx[k];
                    if (in) atomicAdd(&hist[k][bin >> 1], 1u << (16 * (bin & 1u)));
                }
            }
            __syncthreads();
            if (t < 2 * WAVE) {  // one wave per rank
                const int k = t / WAVE;
                const unsigned short* h16 = reinterpret_cast<const unsigned short*>(hist[k]);
                unsigned bin;
                unsigned long long rk;
                bool mine;
                if (level == 2) mine = select_bin<L3_BINS, unsigned short>(h16, srank[k], &bin, &rk);
                else mine = select_bin<L1_BINS, unsigned short>(h16, srank[k], &bin, &rk);
                if (mine) {
                    const int bits = level == 0 ? 0 : (level == 1 ? L2_BITS : L3_BITS);
                    spfx[k] = level == 0 ? bin : ((spfx[k] << bits) | bin);
                    srank[k] = (unsigned)rk;
                }
            }
            __syncthreads();
        }
        if (t == 0) {
            double s_all = 0.0;
            for (int i = 0; i < SAMPLE_THREADS / WAVE; ++i) s_all += red[i];
            const double mean = m ? s_all / (double)m : 0.0;
            ws.shift[q] = (mean == mean && fabs(mean) < 1e300) ? mean : 0.0;
            ws.lo[q] = need[0] ? key2f(spfx[0]) : __int_as_float(0xff800000);
            ws.hi[q] = need[1] ? key2f(spfx[1]) : __int_as_float(0x7f800000);
        }
        __syncthreads();
    }
}

// Unused dynamic LDS per wave of the streaming pass: it only lowers the number of resident waves per CU (8 KB of queues per wave
// = 20 waves per CU otherwise).
static size_t norm_stream_lds_pad() { return (size_t)HK_NORM_LDS_PAD_DEFAULT; }

// Waves per band of the streaming pass: a function of the block SHAPE only (never of the batch size), so a block's
// statistics are the same bits whichever launch it travels in; >= 16 chunks of 1 KB per raster per wave.
__host__ __device__ static inline int pass_waves(int height, int width) {
    const long long chunks = (long long)height * (((width + PX - 1) / PX + WAVE - 1) / WAVE);
    long long w = chunks / 16;
    w = w < 64 ? 64 : (w > PASS_WAVES ? PASS_WAVES : w);
    return (int)w;
}

// Every lane keeps the values it finds between the pivots in its OWN LDS queue (QCAP slots per raster): the value is
// written unconditionally to the slot behind the queue's end and the end moves by the compare result, so the loop body
// has no branch per pixel -- one wave-uniform test per 1 KB chunk asks whether any queue could overflow in the next chunk.
// The order of the compacted values is irrelevant (the select is a histogram).
#ifndef HK_NORM_QCAP
#define HK_NORM_QCAP 16
#endif
constexpr int QCAP = HK_NORM_QCAP;  // (8 slots = 4 KB of LDS per wave was measured slower, alone and beside other streams' kernels)

// DENSE: neither raster has a nodata value (no validity test at all); otherwise the test is branch-free: `cmp` is the
// numeric nodata value or NaN (never equal), `nan` says that NaN is the nodata value.
template <bool DENSE>
__global__ void __launch_bounds__(WAVE) norm_stream_kernel(const NormArgs a, NormWS* __restrict__ ws_all,
                                                            float* __restrict__ mid_all, size_t mid_cap) {
    __shared__ float stage[2][QCAP][WAVE];
    const int band = blockIdx.y, wave = blockIdx.x, lane = threadIdx.x;
    NormWS& ws = ws_all[band];
    const PlaneRef pl = plane_of(a, band);
    // the plane's own wave count (the grid is sized for the largest plane of a batched launch): a plane's partial sums are the
    // same bits whichever launch it travels in
    const int G = pass_waves(pl.height, pl.width);
    if (wave >= G) return;
    const float* __restrict__ sp = pl.sp;
    const float* __restrict__ rp = pl.rp;
    float* mid[2] = {mid_all + ((size_t)band * 2 + 0) * mid_cap, mid_all + ((size_t)band * 2 + 1) * mid_cap};
    const float lo[2] = {ws.lo[0], ws.lo[1]}, hi[2] = {ws.hi[0], ws.hi[1]};
    const double shift[2] = {ws.shift[0], ws.shift[1]};

    unsigned n = 0, le_hi[2] = {0, 0}, n_in[2] = {0, 0};  // per lane: < 2^32 for any block
    double m1[2] = {0.0, 0.0}, m2[2] = {0.0, 0.0};
    unsigned cnt[2] = {0, 0};  // per-lane queue lengths

    auto flush = [&](int q) {
        unsigned incl = cnt[q];
#pragma unroll
        for (int d = 1; d < WAVE; d <<= 1) {
            const unsigned v = __shfl_up(incl, d, WAVE);
            if (lane >= d) incl += v;
        }
        const unsigned total = __shfl(incl, WAVE - 1);
        unsigned base = 0;
        if (lane == 0) base = atomicAdd(&ws.mid_count[q], total);
        base = __shfl(base, 0);
        const size_t off = (size_t)base + (incl - cnt[q]);
        for (unsigned j = 0; j < cnt[q]; ++j)
            if (off + j < mid_cap) mid[q][off + j] = stage[q][j][lane];
        n_in[q] += cnt[q];
        cnt[q] = 0;
    };

    // the wave's chunks: it = wave, wave + G, ... over rows x chunks-per-row, (y, c) stepped without a division
    const int wq = (pl.width + PX - 1) / PX, cpr = (wq + WAVE - 1) / WAVE;
    const long long total = (long long)pl.height * cpr;
    const int dy = G / cpr, dc = G % cpr;
    int py = wave / cpr, pc = wave % cpr;  // position of the next load
    long long pit = wave;
    auto fetch = [&](float4& s4, float4& r4, int& x) {
        // always a valid address (chunk 0 of row 0 for lanes / iterations past the end), x = width marks "no pixels".
        // The row / chunk part of the address is wave-uniform (scalar registers), the lane adds 16 bytes * lane.
        const bool live = pit < total;
        const long long o = live ? (long long)py * pl.stride + (long long)pc * (WAVE * PX) : 0ll;
        const bool ok = live && pc * WAVE + lane < wq;  // rows are padded to a multiple of PX elements
        const int lo4 = ok ? lane * PX : 0;
        // read once: non-temporal, so the pass does not push the rows a tall fit kernel of another stream is about to
        // re-read out of L2 / the memory-side cache (configs[3], four streams: -4 %)
        // (explicitly global: a plane pointer that came out of a batched launch's plane table carries no address space the compiler
        // could infer, and these were flat_load instructions -- which count on the LDS counter too, so that every wait for the
        // compaction queues waited for the stream as well -- until round 6)
        typedef float f4v __attribute__((ext_vector_type(4)));
        typedef __attribute__((address_space(1))) const f4v gf4v;
        const f4v sv = __builtin_nontemporal_load((gf4v*)(sp + o + lo4));
        const f4v rv = __builtin_nontemporal_load((gf4v*)(rp + o + lo4));
        s4 = make_float4(sv.x, sv.y, sv.z, sv.w), r4 = make_float4(rv.x, rv.y, rv.z, rv.w);
        x = ok ? (pc * WAVE + lane) * PX : pl.width;
        pit += G, py += dy, pc += dc;
        if (pc >= cpr) pc -= cpr, ++py;
    };
    const float cmp_s = a.src_nd_mode == 2 ? a.src_nodata : __int_as_float(0x7fc00000);
    const float cmp_r = a.ref_nd_mode == 2 ? a.ref_nodata : __int_as_float(0x7fc00000);
    const bool nan_s = a.src_nd_mode == 1, nan_r = a.ref_nd_mode == 1;
    // FULL (wave-uniform): every lane holds four pixels of the row -- with DENSE the body is then straight-line code
    auto process = [&](const float4& s4, const float4& r4, int x, auto full) {
        constexpr bool FULL = decltype(full)::value;
        if (__any((cnt[0] > cnt[1] ? cnt[0] : cnt[1]) > (unsigned)(QCAP - PX))) {  // rare
            if (__any(cnt[0] > (unsigned)(QCAP - PX))) flush(0);
            if (__any(cnt[1] > (unsigned)(QCAP - PX))) flush(1);
        }
        const float s[PX] = {s4.x, s4.y, s4.z, s4.w}, r[PX] = {r4.x, r4.y, r4.z, r4.w};
        const int npx = pl.width - x;  // <= 0: nothing
#pragma unroll
        for (int i = 0; i < PX; ++i) {
            bool m = FULL || i < npx;
            if constexpr (!DENSE)
                m = m && !(s[i] == cmp_s) && !(r[i] == cmp_r) && (s[i] == s[i] || !nan_s) && (r[i] == r[i] || !nan_r);
            const float v[2] = {s[i], r[i]};
            if (m) {
                ++n;
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    if constexpr ((HK_NORM_ABLATE & 1) != 0) { m1[q] += (double)(v[q] == 1.2345f); continue; }
                    const double d = (double)v[q] - shift[q];
                    m1[q] += d;
                    m2[q] = __fma_rn(d, d, m2[q]);
                }
            }
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                if constexpr ((HK_NORM_ABLATE & 2) != 0) { le_hi[q] += (m && v[q] <= hi[q]) ? 1u : 0u; continue; }
                // two compares per value: "<= hi" is counted, "in [lo, hi]" moves the queue's end; below = le_hi - in
                const bool le = m && v[q] <= hi[q], in = le && !(v[q] < lo[q]);
                stage[q][cnt[q]][lane] = v[q];  // slot cnt <= QCAP - 1 by the test above
                le_hi[q] += le ? 1u : 0u;
                cnt[q] += in ? 1u : 0u;
            }
        }
    };
    const bool quads = (pl.width % PX) == 0;
    auto process_any = [&](const float4& s4, const float4& r4, int x) {
        // wave-uniform: lane 63 holds a whole quad <=> all lanes do
        if (quads && __shfl(x, WAVE - 1) < pl.width) process(s4, r4, x, std::true_type{});
        else process(s4, r4, x, std::false_type{});
    };

    // HK_NORM_SETS register sets in rotation (no copies between them: a copy would wait for the load it copies), so
    // HK_NORM_SETS - 1 chunks per raster are in flight while one is processed
#ifndef HK_NORM_SETS
#define HK_NORM_SETS 3
#endif
    constexpr int NS = HK_NORM_SETS;
    float4 sQ[NS], rQ[NS];
    int xQ[NS];
#pragma unroll
    for (int k = 0; k < NS; ++k) fetch(sQ[k], rQ[k], xQ[k]);
    for (long long it = wave; it < total;) {
        bool more = true;
#pragma unroll
        for (int k = 0; k < NS; ++k) {
            if (more) {
                process_any(sQ[k], rQ[k], xQ[k]);
                fetch(sQ[k], rQ[k], xQ[k]);
                more = (it += G) < total;
            }
        }
    }
    flush(0);
    flush(1);
    const unsigned long long nw = wave_sum((unsigned long long)n);
    unsigned long long bw[2];
    for (int q = 0; q < 2; ++q) {
        m1[q] = wave_sum(m1[q]);
        m2[q] = wave_sum(m2[q]);
        bw[q] = wave_sum((unsigned long long)(le_hi[q] - n_in[q]));  // values below the low pivot
    }
    if (lane == 0) {
        ws.pn[wave] = nw;
        for (int q = 0; q < 2; ++q) ws.p1[q][wave] = m1[q], ws.p2[q][wave] = m2[q], ws.pbelow[q][wave] = bw[q];
    }
}

// statistics + ranks + decision whether the compacted buffers bracket both ranks
constexpr int STATS_THREADS = 256;
__global__ void __launch_bounds__(STATS_THREADS) norm_stats_kernel(NormWS* __restrict__ ws_all, double* __restrict__ norm_out,
                                                                    size_t mid_cap) {
    static_assert(PASS_WAVES % STATS_THREADS == 0, "every thread sums the same number of partials");
    __shared__ unsigned long long sn[STATS_THREADS / WAVE], sb[2][STATS_THREADS / WAVE];
    __shared__ double s1[2][STATS_THREADS / WAVE], s2[2][STATS_THREADS / WAVE];
    NormWS& ws = ws_all[blockIdx.x];
    // every thread sums its stride-256 subset of the per-wave partials in index order (independent loads, issued together),
    // a fixed butterfly inside each wave, then thread 0 adds the four wave sums in order: deterministic
    unsigned long long n = 0, below[2] = {0, 0};
    double m1[2] = {0.0, 0.0}, m2[2] = {0.0, 0.0};
#pragma unroll
    for (int j = 0; j < PASS_WAVES / STATS_THREADS; ++j) {
        const int i = threadIdx.x + j * STATS_THREADS;
        n += ws.pn[i];
#pragma unroll
        for (int q = 0; q < 2; ++q) m1[q] += ws.p1[q][i], m2[q] += ws.p2[q][i], below[q] += ws.pbelow[q][i];
    }
    n = wave_sum(n);
    for (int q = 0; q < 2; ++q) m1[q] = wave_sum(m1[q]), m2[q] = wave_sum(m2[q]), below[q] = wave_sum(below[q]);
    const int w = threadIdx.x / WAVE;
    if ((threadIdx.x & (WAVE - 1)) == 0) {
        sn[w] = n;
        for (int q = 0; q < 2; ++q) s1[q][w] = m1[q], s2[q][w] = m2[q], sb[q][w] = below[q];
    }
    __syncthreads();
    if (threadIdx.x != 0) return;
    n = 0;
    for (int q = 0; q < 2; ++q) m1[q] = m2[q] = 0.0, below[q] = 0;
    for (int i = 0; i < STATS_THREADS / WAVE; ++i) {
        n += sn[i];
        for (int q = 0; q < 2; ++q) m1[q] += s1[q][i], m2[q] += s2[q][i], below[q] += sb[q][i];
    }
    ws.n = n;
    if (n == 0) {  // kernel_model.py:223-226
        norm_out[2 * blockIdx.x] = 0.0;
        norm_out[2 * blockIdx.x + 1] = 0.0;
        ws.done = 1;
        return;
    }
    for (int q = 0; q < 2; ++q) {
        const double d = m1[q] / (double)n;
        ws.mean[q] = ws.shift[q] + d;
        const double v = m2[q] / (double)n - d * d;
        ws.var[q] = v > 0.0 ? v : 0.0;
    }
    const double vi = 0.01 * (double)(n - 1);
    const unsigned long long k0 = (unsigned long long)floor(vi);
    ws.frac = vi - (double)k0;
    ws.k[0] = k0;
    ws.k[1] = k0 + 1 < n ? k0 + 1 : n - 1;
    int fb = 0;
    for (int q = 0; q < 2; ++q) {
        const unsigned long long nm = ws.mid_count[q];
        if (nm > mid_cap || ws.k[0] < below[q] || ws.k[1] >= below[q] + nm) fb = 1;
        ws.sel[q][0].rank = ws.k[0] - below[q];
        ws.sel[q][1].rank = ws.k[1] - below[q];
        ws.sel[q][0].prefix = ws.sel[q][1].prefix = 0;
        // every compacted value has its key in [f2key(lo), f2key(hi)]: stretch that window over the 32 key bits
        const unsigned kb = f2key(ws.lo[q]), range = f2key(ws.hi[q]) - kb;
        ws.kbase[q] = kb;
        ws.ksh[q] = range ? __clz((int)range) : 0;
    }
    if (fb)
        for (int q = 0; q < 2; ++q) {
            ws.sel[q][0].rank = ws.k[0], ws.sel[q][1].rank = ws.k[1];
            ws.kbase[q] = 0, ws.ksh[q] = 0;  // the full rasters are selected on plain keys
        }
    ws.fallback = fb;
}

// ---------------------------------------------------------------------------------------------------------------------
// 3. radix select: histogram of the next key digit of every value whose higher digits match the rank's prefix
// (at the first level both ranks of a raster see the same histogram: only [k = 0] is filled and both selects read it)
__device__ __forceinline__ void hist_add(unsigned* hist, int level, unsigned key, const unsigned (&pfx)[2]) {
    if (level == 0) {
        atomicAdd(&hist[key >> (32 - L1_BITS)], 1u);
        return;
    }
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        if (level == 1) {
            if ((key >> (32 - L1_BITS)) == pfx[k]) atomicAdd(&hist[k * L2_BINS + ((key >> L3_BITS) & (L2_BINS - 1))], 1u);
        } else {
            if ((key >> L3_BITS) == pfx[k]) atomicAdd(&hist[k * L3_BINS + (key & (L3_BINS - 1))], 1u);
        }
    }
}

// LEVEL 0..2 over the compacted buffers (grid: x = workgroups, y = band * 2 + raster)
// (a band whose pivots missed -- ws.fallback -- is histogrammed over its full rasters by the same launch: the chain of a block's
// statistics is a sequence of dependent launches, and each one it does not need is one less to be placed beside other streams' kernels)
template <int LEVEL>
__global__ void __launch_bounds__(NORM_THREADS) norm_mid_hist_kernel(const NormArgs a, NormWS* __restrict__ ws_all,
                                                                       const float* __restrict__ mid_all, size_t mid_cap) {
    constexpr int NB = LEVEL == 0 ? L1_BINS : (LEVEL == 1 ? L2_BINS : L3_BINS);
    __shared__ unsigned hist[2 * NB];
    const int band = blockIdx.y >> 1, q = blockIdx.y & 1;
    NormWS& ws = ws_all[band];
    if (ws.done) return;
    for (int i = threadIdx.x; i < 2 * NB; i += NORM_THREADS) hist[i] = 0;
    __syncthreads();
    if (ws.fallback) {  // uniform over the workgroup: the select over the full rasters (plain keys)
        const PlaneRef pl = plane_of(a, band);
        const unsigned pfx[2] = {ws.sel[q][0].prefix, ws.sel[q][1].prefix};
        const unsigned kb = ws.kbase[q];
        const int ksh = ws.ksh[q];
        for (int y = blockIdx.x; y < pl.height; y += gridDim.x) {
            const long long row = (long long)y * pl.stride;
            for (int x = threadIdx.x; x < pl.width; x += NORM_THREADS) {
                const float s = plane_at(pl.sp, row + x), r = plane_at(pl.rp, row + x);
                if (nvalid(s, a.src_nd_mode, a.src_nodata) && nvalid(r, a.ref_nd_mode, a.ref_nodata))
                    hist_add(hist, LEVEL, (f2key(q ? r : s) - kb) << ksh, pfx);
            }
        }
        __syncthreads();
        unsigned* gh = LEVEL == 0 ? &ws.hist1[q][0][0] : (LEVEL == 1 ? &ws.hist2[q][0][0] : &ws.hist3[q][0][0]);
        for (int i = threadIdx.x; i < 2 * NB; i += NORM_THREADS)
            if (hist[i]) atomicAdd(gh + i, hist[i]);
        return;
    }
    const float* __restrict__ buf = mid_all + ((size_t)band * 2 + q) * mid_cap;
    const unsigned cnt = ws.mid_count[q];
    const unsigned pfx[2] = {ws.sel[q][0].prefix, ws.sel[q][1].prefix};
    // 16-byte loads (the buffers are 256-byte aligned), MID_BLOCKS workgroups per buffer: the pass is bound by the loads in
    // flight.  The keys are taken relative to the pivot window (NormWS::kbase / ksh): the first digit spreads over all bins
    // (on plain keys nearly all compacted values shared one top digit and the LDS atomics serialised on it).
    const unsigned kb = ws.kbase[q];
    const int ksh = ws.ksh[q];
    auto rel = [&](float v) { return (f2key(v) - kb) << ksh; };
    const float4* __restrict__ buf4 = reinterpret_cast<const float4*>(buf);
    const unsigned cnt4 = cnt >> 2;
    for (unsigned i = blockIdx.x * NORM_THREADS + threadIdx.x; i < cnt4; i += gridDim.x * NORM_THREADS) {
        const float4 v = buf4[i];
        hist_add(hist, LEVEL, rel(v.x), pfx), hist_add(hist, LEVEL, rel(v.y), pfx);
        hist_add(hist, LEVEL, rel(v.z), pfx), hist_add(hist, LEVEL, rel(v.w), pfx);
    }
    if (blockIdx.x == 0 && threadIdx.x < (cnt & 3u)) hist_add(hist, LEVEL, rel(buf[(cnt & ~3u) + threadIdx.x]), pfx);
    __syncthreads();
    unsigned* gh = LEVEL == 0 ? &ws.hist1[q][0][0] : (LEVEL == 1 ? &ws.hist2[q][0][0] : &ws.hist3[q][0][0]);
    for (int i = threadIdx.x; i < 2 * NB; i += NORM_THREADS)
        if (hist[i]) atomicAdd(gh + i, hist[i]);  // integer atomics: order-independent result
}

// Fallback: the same select over the full rasters (only bands whose pivots missed).  grid z = raster: 16 KB of LDS per
// workgroup -- these kernels are launched for every block and exit at once in the usual case, but a workgroup has to be
// PLACED before it can exit, and with 32 KB each they queued behind the fit kernels of the other streams.
template <int LEVEL>
__global__ void __launch_bounds__(NORM_THREADS) norm_full_hist_kernel(const NormArgs a, NormWS* __restrict__ ws_all) {
    constexpr int NB = LEVEL == 0 ? L1_BINS : (LEVEL == 1 ? L2_BINS : L3_BINS);
    __shared__ unsigned hist[2 * NB];
    const int band = blockIdx.y, q = blockIdx.z;
    NormWS& ws = ws_all[band];
    if (ws.done || !ws.fallback) return;
    for (int i = threadIdx.x; i < 2 * NB; i += NORM_THREADS) hist[i] = 0;
    __syncthreads();
    const PlaneRef pl = plane_of(a, band);
    const float* __restrict__ sp = pl.sp;
    const float* __restrict__ rp = pl.rp;
    const unsigned pfx[2] = {ws.sel[q][0].prefix, ws.sel[q][1].prefix};
    const unsigned kb = ws.kbase[q];  // 0 / 0 on this path (plain keys)
    const int ksh = ws.ksh[q];
    for (int y = blockIdx.x; y < pl.height; y += gridDim.x) {
        const long long row = (long long)y * pl.stride;
        for (int x = threadIdx.x; x < pl.width; x += NORM_THREADS) {
            const float s = plane_at(sp, row + x), r = plane_at(rp, row + x);
            if (nvalid(s, a.src_nd_mode, a.src_nodata) && nvalid(r, a.ref_nd_mode, a.ref_nodata))
                hist_add(hist, LEVEL, (f2key(q ? r : s) - kb) << ksh, pfx);
        }
    }
    __syncthreads();
    unsigned* gh = LEVEL == 0 ? &ws.hist1[q][0][0] : (LEVEL == 1 ? &ws.hist2[q][0][0] : &ws.hist3[q][0][0]);
    for (int i = threadIdx.x; i < 2 * NB; i += NORM_THREADS)
        if (hist[i]) atomicAdd(gh + i, hist[i]);
}

// After each histogram level (of either path): fix the next digit of the four (raster, rank) keys -- one wave each; after
// the last level the keys are the exact order statistics -> norm.
template <int LEVEL>
__global__ void __launch_bounds__(NORM_THREADS) norm_select_kernel(NormWS* __restrict__ ws_all, double* __restrict__ norm_out) {
    static_assert(NORM_THREADS == 4 * WAVE, "one wave per (raster, rank)");
    NormWS& ws = ws_all[blockIdx.x];
    if (ws.done) return;
    {
        const int q = threadIdx.x >> 7, k = (threadIdx.x >> 6) & 1;
        // the first level's histogram is the same for both ranks: only [k = 0] is filled (hist_add)
        const unsigned* h = LEVEL == 0 ? ws.hist1[q][0] : (LEVEL == 1 ? ws.hist2[q][k] : ws.hist3[q][k]);
        constexpr int NB = LEVEL == 0 ? L1_BINS : (LEVEL == 1 ? L2_BINS : L3_BINS);
        unsigned bin;
        unsigned long long rk;
        if (select_bin<NB>(h, ws.sel[q][k].rank, &bin, &rk)) {
            const int bits = LEVEL == 0 ? 0 : (LEVEL == 1 ? L2_BITS : L3_BITS);
            ws.sel[q][k].prefix = (LEVEL == 0) ? bin : ((ws.sel[q][k].prefix << bits) | bin);
            ws.sel[q][k].rank = rk;
            if (LEVEL == 2) ws.val[q][k] = key2f((ws.sel[q][k].prefix >> ws.ksh[q]) + ws.kbase[q]);
        }
    }
    __syncthreads();
    if (LEVEL == 2 && threadIdx.x == 0) {
        const double n0 = sqrt(ws.var[1]) / sqrt(ws.var[0]);
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

// ---------------------------------------------------------------------------------------------------------------------
// Split blocks: one block's pixels spread over several ranks (each holds some rows), one result.  Every statistic above
// is a sum over pixels -- the shifted moments and the integer histograms of the radix select -- so the ranks run the same
// kernels on their slabs and all-reduce (SUM) a small float64 exchange buffer between the phases:
//   phase 0  sample -> the slab's shift                     | exchange: 2 values per band (the ranks' mean becomes the shift)
//   phase 1  moments of the slab about the common shift     | 5 values per band (n, m1, m2 of src and ref)
//   phase 2  global n, mean, var, ranks; level-1 histogram  | 4 x 2048 counts per band (exact in float64)
//   phase 3  level-1 digit; level-2 histogram               | 4 x 2048
//   phase 4  level-2 digit; level-3 histogram               | 4 x 1024
//   phase 5  level-3 digit -> norm (identical on every rank)
// The select runs over the full slabs (the path the single-GPU code keeps as its fallback): no pivots to agree on.
constexpr int SPLIT_XCHG = 4 * L1_BINS;  // float64 values per band in the exchange buffer

size_t norm_split_exchange_doubles(int n_bands) { return (size_t)n_bands * SPLIT_XCHG; }

__global__ void split_put_shift_kernel(NormWS* __restrict__ ws_all, double* __restrict__ xchg) {
    NormWS& ws = ws_all[blockIdx.x];
    double* x = xchg + (size_t)blockIdx.x * SPLIT_XCHG;
    for (int i = threadIdx.x; i < SPLIT_XCHG; i += blockDim.x) x[i] = i < 2 ? ws.shift[i] : 0.0;
}
__global__ void split_get_shift_kernel(NormWS* __restrict__ ws_all, const double* __restrict__ xchg, double scale) {
    if (threadIdx.x != 0) return;
    NormWS& ws = ws_all[blockIdx.x];
    const double* x = xchg + (size_t)blockIdx.x * SPLIT_XCHG;
    for (int q = 0; q < 2; ++q) {
        ws.shift[q] = x[q] * scale;
        ws.lo[q] = ws.hi[q] = __int_as_float(0xff800000);  // nothing lies between the pivots: the pass takes the moments only
    }
}
__global__ void split_put_moments_kernel(NormWS* __restrict__ ws_all, double* __restrict__ xchg) {
    NormWS& ws = ws_all[blockIdx.x];
    unsigned long long n = 0;
    double m1[2] = {0.0, 0.0}, m2[2] = {0.0, 0.0};
    for (int i = threadIdx.x; i < PASS_WAVES; i += WAVE) {  // fixed order: deterministic
        n += ws.pn[i];
        for (int q = 0; q < 2; ++q) m1[q] += ws.p1[q][i], m2[q] += ws.p2[q][i];
    }
    n = wave_sum(n);
    for (int q = 0; q < 2; ++q) m1[q] = wave_sum(m1[q]), m2[q] = wave_sum(m2[q]);
    if (threadIdx.x != 0) return;
    double* x = xchg + (size_t)blockIdx.x * SPLIT_XCHG;
    x[0] = (double)n, x[1] = m1[0], x[2] = m2[0], x[3] = m1[1], x[4] = m2[1];
}
__global__ void split_get_moments_kernel(NormWS* __restrict__ ws_all, const double* __restrict__ xchg,
                                         double* __restrict__ norm_out) {
    if (threadIdx.x != 0) return;
    NormWS& ws = ws_all[blockIdx.x];
    const double* x = xchg + (size_t)blockIdx.x * SPLIT_XCHG;
    const unsigned long long n = (unsigned long long)x[0];
    ws.n = n;
    ws.mid_count[0] = ws.mid_count[1] = 0;
    if (n == 0) {  // kernel_model.py:223-226
        norm_out[2 * blockIdx.x] = 0.0, norm_out[2 * blockIdx.x + 1] = 0.0;
        ws.done = 1;
        return;
    }
    if (n >= (1ull << 32)) {
        // the select's histogram bins are 32-bit: the all-reduced count of one bin could wrap for a block of 2^32 valid
        // pixels or more (byte imagery lands in a handful of bins).  Refuse it identically on every rank (x[0] is the
        // all-reduced count) instead of returning wrong percentiles: NaN statistics.
        const double qn = __longlong_as_double(0x7ff8000000000000ll);
        norm_out[2 * blockIdx.x] = qn, norm_out[2 * blockIdx.x + 1] = qn;
        ws.done = 1;
        return;
    }
    for (int q = 0; q < 2; ++q) {
        const double d = x[1 + 2 * q] / (double)n;
        ws.mean[q] = ws.shift[q] + d;
        const double v = x[2 + 2 * q] / (double)n - d * d;
        ws.var[q] = v > 0.0 ? v : 0.0;
    }
    const double vi = 0.01 * (double)(n - 1);
    const unsigned long long k0 = (unsigned long long)floor(vi);
    ws.frac = vi - (double)k0;
    ws.k[0] = k0, ws.k[1] = k0 + 1 < n ? k0 + 1 : n - 1;
    for (int q = 0; q < 2; ++q)
        for (int k = 0; k < 2; ++k) ws.sel[q][k].rank = ws.k[k], ws.sel[q][k].prefix = 0;
    ws.fallback = 1;  // the select runs over the full slabs
}
template <int LEVEL>
__global__ void split_hist_kernel(NormWS* __restrict__ ws_all, double* __restrict__ xchg, int put) {
    constexpr int NB = LEVEL == 0 ? L1_BINS : (LEVEL == 1 ? L2_BINS : L3_BINS);
    NormWS& ws = ws_all[blockIdx.x];
    unsigned* h = LEVEL == 0 ? &ws.hist1[0][0][0] : (LEVEL == 1 ? &ws.hist2[0][0][0] : &ws.hist3[0][0][0]);
    double* x = xchg + (size_t)blockIdx.x * SPLIT_XCHG;
    for (int i = threadIdx.x; i < SPLIT_XCHG; i += blockDim.x) {
        if (put) x[i] = i < 4 * NB ? (double)h[i] : 0.0;
        else if (i < 4 * NB) h[i] = (unsigned)x[i];
    }
}

// One phase of the split-block statistics on this rank's slab (`a`); the caller all-reduces `xchg` between the phases.
// `inv_world` = 1 / number of ranks (phase 1 turns the sum of the slabs' shifts into their mean).
hipError_t launch_block_norm_split(const NormArgs& a, void* workspace, double* xchg, double inv_world, int phase,
                                   double* norm_out, hipStream_t stream) {
    NormWS* ws = reinterpret_cast<NormWS*>(workspace);
    const dim3 bands(a.n_bands), block(NORM_THREADS), gfull(FB_BLOCKS, a.n_bands, 2);
    // a rank whose slab has no rows takes part with zeros: it skips the kernels that read pixels, not the exchange
    const bool rows = a.height > 0;
    switch (phase) {
    case 0: {
        hipError_t e = hipMemsetAsync(ws, 0, sizeof(NormWS) * (size_t)a.n_bands, stream);
        if (e != hipSuccess) return e;
        if (rows) HK_LAUNCH(norm_sample_kernel, bands, dim3(SAMPLE_THREADS), 0, stream, a, ws);
        HK_LAUNCH(split_put_shift_kernel, bands, block, 0, stream, ws, xchg);
        break;
    }
    case 1: {
        HK_LAUNCH(split_get_shift_kernel, bands, dim3(WAVE), 0, stream, ws, xchg, inv_world);
        if (rows) {
            const dim3 gstream(pass_waves(a.height, a.width), a.n_bands);
            float* mid = reinterpret_cast<float*>(static_cast<char*>(workspace) + align256(sizeof(NormWS) * (size_t)a.n_bands));
            const size_t cap_al = align256(mid_capacity((long long)a.height * a.width) * sizeof(float)) / sizeof(float);
            if (a.src_nd_mode == 0 && a.ref_nd_mode == 0)
                HK_LAUNCH(norm_stream_kernel<true>, gstream, dim3(WAVE), norm_stream_lds_pad(), stream, a, ws, mid, cap_al);
            else
                HK_LAUNCH(norm_stream_kernel<false>, gstream, dim3(WAVE), norm_stream_lds_pad(), stream, a, ws, mid, cap_al);
        }
        HK_LAUNCH(split_put_moments_kernel, bands, dim3(WAVE), 0, stream, ws, xchg);
        break;
    }
    case 2:
        HK_LAUNCH(split_get_moments_kernel, bands, dim3(WAVE), 0, stream, ws, xchg, norm_out);
        if (rows) HK_LAUNCH(norm_full_hist_kernel<0>, gfull, block, 0, stream, a, ws);
        HK_LAUNCH(split_hist_kernel<0>, bands, block, 0, stream, ws, xchg, 1);
        break;
    case 3:
        HK_LAUNCH(split_hist_kernel<0>, bands, block, 0, stream, ws, xchg, 0);
        HK_LAUNCH(norm_select_kernel<0>, bands, block, 0, stream, ws, norm_out);
        if (rows) HK_LAUNCH(norm_full_hist_kernel<1>, gfull, block, 0, stream, a, ws);
        HK_LAUNCH(split_hist_kernel<1>, bands, block, 0, stream, ws, xchg, 1);
        break;
    case 4:
        HK_LAUNCH(split_hist_kernel<1>, bands, block, 0, stream, ws, xchg, 0);
        HK_LAUNCH(norm_select_kernel<1>, bands, block, 0, stream, ws, norm_out);
        if (rows) HK_LAUNCH(norm_full_hist_kernel<2>, gfull, block, 0, stream, a, ws);
        HK_LAUNCH(split_hist_kernel<2>, bands, block, 0, stream, ws, xchg, 1);
        break;
    case 5:
        HK_LAUNCH(split_hist_kernel<2>, bands, block, 0, stream, ws, xchg, 0);
        HK_LAUNCH(norm_select_kernel<2>, bands, block, 0, stream, ws, norm_out);
        break;
    default:
        return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

int norm_pass_waves(int height, int width) { return pass_waves(height, width); }

hipError_t launch_block_norm(const NormArgs& a, void* workspace, double* norm_out, hipStream_t stream) {
    NormWS* ws = reinterpret_cast<NormWS*>(workspace);
    const size_t cap = mid_capacity((long long)a.height * a.width);
    const size_t cap_al = align256(cap * sizeof(float)) / sizeof(float);
    float* mid = reinterpret_cast<float*>(static_cast<char*>(workspace) + align256(sizeof(NormWS) * (size_t)a.n_bands));
    hipError_t e = hipMemsetAsync(ws, 0, sizeof(NormWS) * (size_t)a.n_bands, stream);
    if (e != hipSuccess) return e;
    const dim3 bands(a.n_bands), block(NORM_THREADS);
    HK_LAUNCH(norm_sample_kernel, bands, dim3(SAMPLE_THREADS), 0, stream, a, ws);
    const dim3 gstream(a.grid_waves > 0 ? a.grid_waves : pass_waves(a.height, a.width), a.n_bands);
    if (a.src_nd_mode == 0 && a.ref_nd_mode == 0)
        HK_LAUNCH(norm_stream_kernel<true>, gstream, dim3(WAVE), norm_stream_lds_pad(), stream, a, ws, mid, cap_al);
    else
        HK_LAUNCH(norm_stream_kernel<false>, gstream, dim3(WAVE), norm_stream_lds_pad(), stream, a, ws, mid, cap_al);
    HK_LAUNCH(norm_stats_kernel, bands, dim3(STATS_THREADS), 0, stream, ws, norm_out, cap_al);
    // workgroups per compacted buffer: at least 8 float4 per thread of a full buffer (a 4096^2 block's buffers hold ~0.2 M values:
    // 256 workgroups would spend their time zeroing and merging 16 KB histograms -- 0.40 -> 0.15 ms for configs[3]'s 128 blocks)
    const size_t mid_wgs = cap_al / 4 / (NORM_THREADS * 8);
    // (at least 32 workgroups per buffer: a band that fell back has its full rasters histogrammed by the same grid)
    const dim3 gmid((unsigned)(mid_wgs < 32 ? 32 : (mid_wgs > MID_BLOCKS ? MID_BLOCKS : mid_wgs)), a.n_bands * 2);
    HK_LAUNCH(norm_mid_hist_kernel<0>, gmid, block, 0, stream, a, ws, mid, cap_al);
    HK_LAUNCH(norm_select_kernel<0>, bands, block, 0, stream, ws, norm_out);
    HK_LAUNCH(norm_mid_hist_kernel<1>, gmid, block, 0, stream, a, ws, mid, cap_al);
    HK_LAUNCH(norm_select_kernel<1>, bands, block, 0, stream, ws, norm_out);
    HK_LAUNCH(norm_mid_hist_kernel<2>, gmid, block, 0, stream, a, ws, mid, cap_al);
    HK_LAUNCH(norm_select_kernel<2>, bands, block, 0, stream, ws, norm_out);
    return hipGetLastError();
}

}  // namespace hk
