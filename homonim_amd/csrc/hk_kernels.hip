// hk_kernels.hip -- gfx950 (MI355X / CDNA4) kernels of the homonim kernel-model hot path: dispatch of the fused fit(+apply)
// kernel (template: hk_fit_kernel.h, instantiated per MODEL x R2 in hk_fit_tu.hip), KernelModel.apply alone, the synthetic
// workload, the stream probe and the self-test.
#include "hk_fit_kernel.h"

#include <cstdio>
#include <cstring>
#include <mutex>

namespace hk {

// ---------------------------------------------------------------------------------------------------------------------
// Launch ledger (hk_kernels.h BuildRecord): the list's head and its lock are function-local statics, because the records of the
// other translation units register from their own static initialisers, in no particular order relative to this file's.
static BuildRecord*& ledger_head() {
    static BuildRecord* head = nullptr;
    return head;
}
static std::mutex& ledger_mu() {
    static std::mutex mu;
    return mu;
}
static void ledger_add(BuildRecord* r) {
    std::lock_guard<std::mutex> lk(ledger_mu());
    r->launches = 0;
    r->next = ledger_head();
    ledger_head() = r;
}
BuildRecord::BuildRecord(const char* kernel_name) {
    snprintf(name, sizeof(name), "%s", kernel_name);
    ledger_add(this);
}
BuildRecord::BuildRecord(int model, bool r2, int rw, bool dense, int ring, bool cert_only, int wpb, bool batch) {
    snprintf(name, sizeof(name), "fit_apply_kernel<%d,%d,%d,%d,%d,%d,%d,%d>", model, (int)r2, rw, (int)dense, ring, (int)cert_only, wpb,
             (int)batch);
    ledger_add(this);
}
BuildRecord::BuildRecord(const char* kernel_name, int model, bool r2, int rw, bool dense, int ring) {
    snprintf(name, sizeof(name), "%s<%d,%d,%d,%d,%d>", kernel_name, model, (int)r2, rw, (int)dense, ring);
    ledger_add(this);
}
size_t ledger_text(char* buf, size_t len, bool reset) {
    std::lock_guard<std::mutex> lk(ledger_mu());
    size_t need = 1;
    for (BuildRecord* r = ledger_head(); r; r = r->next) {
        char line[128];
        const unsigned long long n = reset ? __atomic_exchange_n(&r->launches, 0ull, __ATOMIC_RELAXED)
                                           : __atomic_load_n(&r->launches, __ATOMIC_RELAXED);
        const int m = snprintf(line, sizeof(line), "%s\t%llu\n", r->name, n);
        if (buf && need + (size_t)m <= len) memcpy(buf + need - 1, line, (size_t)m);
        need += (size_t)m;
    }
    if (buf && len) buf[(need <= len ? need : len) - 1] = '\0';
    return need;
}


#ifdef HK_FIT_ONE_TU  // A/B tooling (tools/mkvariant*.sh): every instantiation in this translation unit
hipError_t launch_fit_m0_r0(const FitArgs& a, hipStream_t stream) { return launch_dense<0, false>(a, stream); }
hipError_t launch_fit_m0_r1(const FitArgs& a, hipStream_t stream) { return launch_dense<0, true>(a, stream); }
hipError_t launch_fit_m1_r0(const FitArgs& a, hipStream_t stream) { return launch_dense<1, false>(a, stream); }
hipError_t launch_fit_m1_r1(const FitArgs& a, hipStream_t stream) { return launch_dense<1, true>(a, stream); }
hipError_t launch_fit_m2_r0(const FitArgs& a, hipStream_t stream) { return launch_dense<2, false>(a, stream); }
hipError_t launch_fit_m2_r1(const FitArgs& a, hipStream_t stream) { return launch_dense<2, true>(a, stream); }
hipError_t read_stamps(unsigned long long* out16, bool reset) {
    for (int k = 0; k < 16; ++k) out16[k] = 0;
    return read_stamps_tu(out16, reset);
}
#else
hipError_t read_stamps(unsigned long long* out16, bool reset) {
    for (int k = 0; k < 16; ++k) out16[k] = 0;
    hipError_t (*const tus[6])(unsigned long long*, bool) = {read_stamps_m0_r0, read_stamps_m0_r1, read_stamps_m1_r0,
                                                              read_stamps_m1_r1, read_stamps_m2_r0, read_stamps_m2_r1};
    for (auto f : tus) {
        const hipError_t e = f(out16, reset);
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}
#endif

hipError_t launch_fit_apply(const FitArgs& a, int model, bool with_r2, hipStream_t stream) {
    switch (model * 2 + (with_r2 ? 1 : 0)) {
        case 0: return launch_fit_m0_r0(a, stream);
        case 1: return launch_fit_m0_r1(a, stream);
        case 2: return launch_fit_m1_r0(a, stream);
        case 3: return launch_fit_m1_r1(a, stream);
        case 4: return launch_fit_m2_r0(a, stream);
        case 5: return launch_fit_m2_r1(a, stream);
    }
    return hipErrorInvalidValue;
}

int fit_lockstep_waves() { return HK_WPB_MEM; }
bool fit_batch_supported(int model, bool with_r2) { return fit_batch_build(model, with_r2); }
size_t fit_lds_bytes(int kh, int ring_mode, bool ahead) { return fit_lds_bytes_of(kh, ring_mode, ahead); }

// ---------------------------------------------------------------------------------------------------------------------
// KernelModel.apply alone (kernel_model.py:461): used after parameters were re-sampled / in-painted on another grid.
__global__ void __launch_bounds__(256) apply_kernel(const float* __restrict__ src, const float* __restrict__ gain,
                                                    const float* __restrict__ offset, float* __restrict__ out,
                                                    int height, int width, long long stride) {
    const int x = (blockIdx.x * blockDim.x + threadIdx.x) * PX;
    if (x >= width) return;
    for (int y = blockIdx.y; y < height; y += gridDim.y) {
        const long long off = (long long)y * stride + x;
        if (x + PX <= width) {
            const float4 s = *reinterpret_cast<const float4*>(src + off);
            const float4 g = *reinterpret_cast<const float4*>(gain + off);
            const float4 o = *reinterpret_cast<const float4*>(offset + off);
            float4 c;
            c.x = __fadd_rn(__fmul_rn(g.x, s.x), o.x);
            c.y = __fadd_rn(__fmul_rn(g.y, s.y), o.y);
            c.z = __fadd_rn(__fmul_rn(g.z, s.z), o.z);
            c.w = __fadd_rn(__fmul_rn(g.w, s.w), o.w);
            *reinterpret_cast<float4*>(out + off) = c;
        } else {
            for (int i = 0; x + i < width; ++i) out[off + i] = __fadd_rn(__fmul_rn(gain[off + i], src[off + i]), offset[off + i]);
        }
    }
}

hipError_t launch_apply(const float* src, const float* gain, const float* offset, float* out, int height, int width,
                        long long stride, hipStream_t stream) {
    const int threads = 256;
    const int gx = (width + threads * PX - 1) / (threads * PX);
    const int gy = height < 4096 ? height : 4096;
    HK_LAUNCH(apply_kernel, dim3(gx, gy), dim3(threads), 0, stream, src, gain, offset, out, height, width,
                       stride);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------------
// Synthetic workload, generated in place in HBM: counter-based hash RNG, so any tile of the
// raster can be regenerated independently.  Test / bench data only.
__device__ __forceinline__ unsigned long long mix64(unsigned long long z) {
    z += 0x9e3779b97f4a7c15ull;
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return z ^ (z >> 31);
}
__device__ __forceinline__ float u01(unsigned long long h) { return (float)(h >> 40) * (1.0f / 16777216.0f); }

__global__ void __launch_bounds__(256) synth_kernel(float* __restrict__ src, float* __restrict__ ref, int n_bands,
                                                    int height, int width, long long stride, long long band_stride,
                                                    unsigned long long seed, int nodata_variant) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int band = blockIdx.z;
    if (x >= width) return;
    for (int y = blockIdx.y; y < height; y += gridDim.y) {
        const unsigned long long key = ((unsigned long long)band << 56) ^ ((unsigned long long)y << 28) ^ (unsigned long long)x;
        const unsigned long long h0 = mix64(key ^ mix64(seed));
        const unsigned long long h1 = mix64(h0), h2 = mix64(h1), h3 = mix64(h2);
        const float s = 0.05f + 0.95f * u01(h0);
        const float u1 = fmaxf(u01(h1), 5.9604645e-8f), u2 = u01(h2);
        const float z = sqrtf(-2.0f * __logf(u1)) * __cosf(6.2831853f * u2);
        const float g = 1.2f + 0.3f * __sinf((float)x / 97.f) * __cosf((float)y / 131.f);
        const float o = 0.05f * (1.f + 0.5f * __sinf((float)y / 211.f));
        // 3: noisy reference, R2 around the threshold (35 % of the pixels fail 0.25); 4: very noisy (85 % fail, like real pairs)
        float r = g * s + o + (nodata_variant == 3 ? 0.5f : (nodata_variant == 4 ? 1.5f : 0.01f)) * z;
        float sv = s;
        if (nodata_variant == 5) {
            // measurement aid: LOW-ENTROPY data -- 64 source levels, the reference an exact affine image of them: few bits toggle
            // on the wires and in the ALUs.  The instruction stream is the headline's, the energy per launch is not (FLOOR.md
            // section 2).
            sv = 0.25f + (float)(h0 & 63ull) * 0.0078125f;
            r = 1.25f * sv + 0.125f;
        }
        if (nodata_variant == 1 || nodata_variant == 2 || nodata_variant == 6) {  // 1: NaN frame + 0.1 % holes, 2: NaN frame only
            const bool frame = x < 3 || y < 3 || x >= width - 3 || y >= height - 3;
            const bool holes = nodata_variant == 1;
            if (frame || (holes && (h3 & 0xffffu) < 66u)) sv = qnan();            // ~0.1 %
            if (frame || (holes && ((h3 >> 16) & 0xffffu) < 66u)) r = qnan();
            if (nodata_variant == 6) {
                // 6: NaN frame + ~1 % of the area in round blobs 32 - 128 pixels across, source and reference independently -- what
                // cloud / shadow masks look like (single-pixel holes at random are a worst case no sensor produces).  Every second
                // 512 x 512 cell owns one blob (centre and radius hashed from the cell); a pixel looks at its cell and the 8 around it.
                for (int salt = 0; salt < 2; ++salt) {
                    bool in = false;
                    for (int dy = -1; dy <= 1; ++dy)
                        for (int dx = -1; dx <= 1; ++dx) {
                            const long long cx = (x >> 9) + dx, cy = (y >> 9) + dy;
                            const unsigned long long hc = mix64(((unsigned long long)band << 58) ^ ((unsigned long long)salt << 56) ^
                                                                ((unsigned long long)(cy + 8) << 28) ^ (unsigned long long)(cx + 8) ^ mix64(seed ^ 0x6b10b5ull));
                            if (hc & 1ull) {
                                const long long bx = (cx << 9) + (long long)((hc >> 8) & 511ull), by = (cy << 9) + (long long)((hc >> 20) & 511ull);
                                const long long rad = 16 + (long long)((hc >> 32) % 49ull), ddx = x - bx, ddy = y - by;
                                in |= ddx * ddx + ddy * ddy <= rad * rad;
                            }
                        }
                    if (in) (salt ? r : sv) = qnan();
                }
            }
        }
        const long long off = (long long)band * band_stride + (long long)y * stride + x;
        src[off] = sv;
        ref[off] = r;
    }
}

hipError_t launch_synth_fill(float* src, float* ref, int n_bands, int height, int width, long long stride,
                             long long band_stride, unsigned long long seed, int nodata_variant, hipStream_t stream) {
    const int threads = 256;
    const int gy = height < 2048 ? height : 2048;
    HK_LAUNCH(synth_kernel, dim3((width + threads - 1) / threads, gy, n_bands), dim3(threads), 0, stream, src,
                       ref, n_bands, height, width, stride, band_stride, seed, nodata_variant);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------------
// Flat 2-read 1-write stream (bench.py `roofline.copy_gbps_measured`; tools/ubench_copy.hip holds the sweep this shape
// came out of: persistent grid of 4 workgroups per CU, four 16-byte loads in flight per lane and array, non-temporal).
__global__ void __launch_bounds__(256) stream_probe_kernel(const hk_v4* __restrict__ a, const hk_v4* __restrict__ b,
                                                           hk_v4* __restrict__ o, size_t n) {
    constexpr int U = 4;
    const size_t chunk = (size_t)U * 256;
    for (size_t base = (size_t)blockIdx.x * chunk; base < n; base += (size_t)gridDim.x * chunk) {
        hk_v4 va[U], vb[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const size_t i = base + (size_t)u * 256 + threadIdx.x;
            va[u] = i < n ? __builtin_nontemporal_load(a + i) : hk_v4{0.f, 0.f, 0.f, 0.f};
            vb[u] = i < n ? __builtin_nontemporal_load(b + i) : hk_v4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const size_t i = base + (size_t)u * 256 + threadIdx.x;
            if (i < n) __builtin_nontemporal_store(va[u] + vb[u], o + i);
        }
    }
}

hipError_t launch_stream_probe(const void* a, const void* b, void* out, size_t n_bytes, hipStream_t stream) {
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    HK_LAUNCH(stream_probe_kernel, dim3(cus * 4), dim3(256), 0, stream, reinterpret_cast<const hk_v4*>(a),
                       reinterpret_cast<const hk_v4*>(b), reinterpret_cast<hk_v4*>(out), n_bytes / 16);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------------
// Exact, order-free checksum of a window of a float32 plane: the sum of its pixels' BIT PATTERNS modulo 2^64 (test / bench aid:
// the union of N ranks' shards against the single-rank result without moving the rasters to the host).
__global__ void __launch_bounds__(256) checksum_kernel(const unsigned* __restrict__ p, long long stride, int height, int width,
                                                       unsigned long long* __restrict__ out) {
    unsigned long long acc = 0ull;
    for (int y = blockIdx.y; y < height; y += gridDim.y)
        for (int x = blockIdx.x * blockDim.x + threadIdx.x; x < width; x += gridDim.x * blockDim.x)
            acc += p[(long long)y * stride + x];
#pragma unroll
    for (int d = WAVE / 2; d > 0; d >>= 1) acc += __shfl_xor(acc, d);
    if ((threadIdx.x & (WAVE - 1)) == 0 && acc) atomicAdd(out, acc);
}

hipError_t launch_checksum(const float* plane, long long stride, int height, int width, unsigned long long* out_dev, hipStream_t stream) {
    const int gx = (width + 255) / 256 < 64 ? (width + 255) / 256 : 64, gy = height < 1024 ? height : 1024;
    HK_LAUNCH(checksum_kernel, dim3(gx, gy), dim3(256), 0, stream, reinterpret_cast<const unsigned*>(plane), stride, height, width, out_dev);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------------
// Self-test of the DPP wave shifts + the compile-time horizontal sums against a brute-force definition.
template <int RW>
__device__ int hsum_check(int lane) {
    double V[PX], Hd[PX];
    int Vi[PX], Hi[PX];
#pragma unroll
    for (int i = 0; i < PX; ++i) {
        Vi[i] = (lane * PX + i) * 3 + 1;
        V[i] = (double)Vi[i] + 0.5;
    }
    hsum<RW, double>(V, Hd, lane);
    hsum<RW, int>(Vi, Hi, lane);
    constexpr int OL = (RW + PX - 1) / PX;
    int bad = 0;
    if (lane >= OL && lane < WAVE - OL) {
#pragma unroll
        for (int i = 0; i < PX; ++i) {
            const int c = lane * PX + i;
            int ei = 0;
            double ed = 0.0;
            for (int d = -RW; d <= RW; ++d) {
                ei += (c + d) * 3 + 1;
                ed += (double)((c + d) * 3 + 1) + 0.5;
            }
            if (Hi[i] != ei || Hd[i] != ed) bad = 1;
        }
    }
    return bad;
}

template <int E>
__device__ int hsum_wide_check(int lane) {
    double V[PX], Hd[PX];
    int Vi[PX], Hi[PX];
#pragma unroll
    for (int i = 0; i < PX; ++i) {
        Vi[i] = (lane * PX + i) * 3 + 1;
        V[i] = (double)Vi[i] + 0.5;
    }
    int bad = 0;
    for (int f = 1; f <= 6; ++f) {
        const int rw = PX * f + E, ol = (rw + PX - 1) / PX;
        const WideLanes wl = make_wide_lanes(rw, lane);
        if (wide_pairs(E, f)) {   // (both forms where the width allows the paired one: the same sums)
            double Hp[PX];
            int Hpi[PX];
            hsum_wide<E, double, true>(V, Hp, wl, lane);
            hsum_wide<E, int, true>(Vi, Hpi, wl, lane);
            if (lane >= ol && lane < WAVE - ol)
                for (int i = 0; i < PX; ++i) {
                    int ei = 0;
                    double ed = 0.0;
                    for (int d = -rw; d <= rw; ++d) ei += (lane * PX + i + d) * 3 + 1, ed += (double)((lane * PX + i + d) * 3 + 1) + 0.5;
                    if (Hpi[i] != ei || Hp[i] != ed) bad = 1;
                }
        }
        hsum_wide<E, double>(V, Hd, wl, lane);
        hsum_wide<E, int>(Vi, Hi, wl, lane);
        if (lane >= ol && lane < WAVE - ol) {
            for (int i = 0; i < PX; ++i) {
                int ei = 0;
                double ed = 0.0;
                for (int d = -rw; d <= rw; ++d) {
                    ei += (lane * PX + i + d) * 3 + 1;
                    ed += (double)((lane * PX + i + d) * 3 + 1) + 0.5;
                }
                if (Hi[i] != ei || Hd[i] != ed) bad = 1;
            }
        }
    }
    return bad;
}

__global__ void selftest_kernel(int* result) {
    const int lane = threadIdx.x;
    int code = 0;
    const int l = dpp_from_left(lane + 100), r = dpp_from_right(lane + 100);
    if (l != (lane == 0 ? 0 : lane + 99)) code |= 1;
    if (r != (lane == WAVE - 1 ? 0 : lane + 101)) code |= 2;
    const double dl = dpp_from_left((double)lane + 0.25);
    if (dl != (lane == 0 ? 0.0 : (double)(lane - 1) + 0.25)) code |= 4;
    if (hsum_check<1>(lane)) code |= 8;
    if (hsum_check<2>(lane)) code |= 16;
    if (hsum_check<3>(lane)) code |= 32;
    if (hsum_check<7>(lane)) code |= 64;
    {   // 15 wide with the distance-2 neighbours through chained DPP shifts
        double V[PX], Hd[PX], Hc[PX];
#pragma unroll
        for (int i = 0; i < PX; ++i) V[i] = (double)((lane * PX + i) * 3 + 1) + 0.5;
        hsum<7, double>(V, Hd, lane);
        hsum<7, double, false, true>(V, Hc, lane);
        if (lane >= 2 && lane < WAVE - 2)
            for (int i = 0; i < PX; ++i)
                if (Hd[i] != Hc[i]) code |= 64;
    }
    if (hsum_check<4>(lane) || hsum_check<5>(lane) || hsum_check<6>(lane)) code |= 256;
    // kernels wider than 15: E = rw mod 4 at compile time, F = rw / 4 at run time (F = 1 .. 6: 9 to 55 wide)
    if (hsum_wide_check<0>(lane) || hsum_wide_check<1>(lane) || hsum_wide_check<2>(lane) || hsum_wide_check<3>(lane)) code |= 128;
    {   // the accuracy fast_quot() assumes of v_rcp_f64 (2^-22; appendix B of DESIGN.md) with a factor 2 in hand
        unsigned long long z = 0x9e3779b97f4a7c15ull * (unsigned long long)(lane + 1);
        for (int it = 0; it < 512; ++it) {
            z ^= z << 13, z ^= z >> 7, z ^= z << 17;
            const double m = 1.0 + (double)(z >> 12) * 0x1p-52;
            const double d = ldexp((z & 1) ? -m : m, (int)((z >> 1) % 201) - 100);
            const double y0 = __builtin_amdgcn_rcp(d);
            const double rel = __fma_rn(-d, y0, 1.0);  // 1 - d*y0 = -eps0 (exact to 2^-53)
            if (!(fabs(rel) < 0x1p-23)) code |= 512;
            // and the guarded quotient against the IEEE one
            const double n = (double)(float)((double)(it + 1) * 0.37 - 90.0);
            const double q = fast_quot(n, d);
            const bool again = (quot_guard(q) < 2u * HK_DIV_GUARD + 1u) | (quot_range(q) > 0x0fd00000u);
            if (!again && (float)q != (float)__ddiv_rn(n, d)) code |= 1024;
        }
    }
    if (code) atomicOr(result, code);
}

hipError_t launch_selftest(int* result_dev, hipStream_t stream) {
    HK_LAUNCH(selftest_kernel, dim3(1), dim3(WAVE), 0, stream, result_dev);
    return hipGetLastError();
}

}  // namespace hk
