// Micro-benchmark: HBM throughput of the fit kernel's ACCESS PATTERN without its arithmetic -- one wave marches down a
// column strip (64 lanes x 16 B per row per input, 62 lanes store), segments of `seg` rows (+ 2*rh priming rows), units
// dealt to XCDs in runs of `remap`.  Variants: waves per workgroup (neighbouring strips in one workgroup), rows in
// flight per wave, non-temporal loads / stores.  A flat float4 copy of the same bytes is the reference line.
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench_strips.hip -o tools/ubench_strips
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

struct Args {
    const float* src; const float* ref; float* out;
    int height, width; long long stride, band_stride; int n_bands;
    int seg, n_strips, n_segs, total_units, remap, rh;
};

template <int WAVES, int DEPTH, bool NT_LD, bool NT_ST, int SYNC = 0>
__global__ void __launch_bounds__(64 * WAVES) strips(const Args a) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    int ublock = blockIdx.x;
    if (a.remap) {
        const int g = a.remap, slot = blockIdx.x >> 3;
        ublock = ((slot / g) * 8 + (blockIdx.x & 7)) * g + slot % g;
    }
    const int unit = ublock * WAVES + w;
    if (unit >= a.total_units && !SYNC) return;  // (the lock-step variants keep every wave: their loads are clamped)
    const int strip = unit % a.n_strips, t0 = unit / a.n_strips, band = t0 % a.n_bands, seg = t0 / a.n_bands;
    const int x = (strip * 62 + lane - 1) * 4;
    const int y0 = seg * a.seg, y1 = min(y0 + a.seg, a.height);
    const bool lane_in = x >= 0 && x < a.width;
    const unsigned xq = lane_in ? (unsigned)x * 4u : 0u;
    const bool out_lane = lane >= 1 && lane < 63 && lane_in;
    const float* sp = a.src + (long long)band * a.band_stride;
    const float* rp = a.ref + (long long)band * a.band_stride;
    float* op = a.out + (long long)band * a.band_stride;
    auto ld = [&](const float* base, int row) {
        const int rc = min(max(row, 0), a.height - 1);
        const float4* p = reinterpret_cast<const float4*>(reinterpret_cast<const char*>(base + (long long)rc * a.stride) + xq);
        if constexpr (NT_LD) {
            typedef float v4 __attribute__((ext_vector_type(4)));
            const v4 v = __builtin_nontemporal_load(reinterpret_cast<const v4*>(p));
            return make_float4(v.x, v.y, v.z, v.w);
        } else {
            return *p;
        }
    };
    float4 qs[DEPTH], qr[DEPTH];
    const int t_first = y0 - a.rh, t_last = y1 - 1 + a.rh;
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) { qs[d] = ld(sp, min(t_first + d, t_last)); qr[d] = ld(rp, min(t_first + d, t_last)); }
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int t = t_first; t <= t_last; ++t) {
        if constexpr (SYNC > 0) {  // the workgroup's waves (adjacent strips) stay within SYNC rows of each other
            if (((t - t_first) & (SYNC - 1)) == 0) __syncthreads();
        }
        const float4 s = qs[0], r = qr[0];
#pragma unroll
        for (int d = 0; d + 1 < DEPTH; ++d) { qs[d] = qs[d + 1]; qr[d] = qr[d + 1]; }
        qs[DEPTH - 1] = ld(sp, min(t + DEPTH, t_last));
        qr[DEPTH - 1] = ld(rp, min(t + DEPTH, t_last));
        acc.x += s.x * r.x; acc.y += s.y * r.y; acc.z += s.z * r.z; acc.w += s.w * r.w;
        const int y = t - a.rh;
        if (y >= y0 && out_lane && unit < a.total_units) {
            float4* p = reinterpret_cast<float4*>(reinterpret_cast<char*>(op + (long long)y * a.stride) + (unsigned)x * 4u);
            const float4 c = make_float4(acc.x + s.x, acc.y + s.y, acc.z + s.z, acc.w + s.w);
            if constexpr (NT_ST) {
                typedef float v4 __attribute__((ext_vector_type(4)));
                __builtin_nontemporal_store(v4{c.x, c.y, c.z, c.w}, reinterpret_cast<v4*>(p));
            } else {
                *p = c;
            }
        }
    }
}

__global__ void __launch_bounds__(256) flat_copy(const float4* __restrict__ s, const float4* __restrict__ r, float4* __restrict__ o, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float4 a = s[i], b = r[i];
        o[i] = make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
    }
}

template <int WAVES, int DEPTH, bool NT_LD, bool NT_ST, int SYNC = 0>
float run(Args a, int reps) {
    a.total_units = a.n_strips * a.n_segs * a.n_bands;
    int blocks = (a.total_units + WAVES - 1) / WAVES;
    if (a.remap) blocks = (blocks + 8 * a.remap - 1) / (8 * a.remap) * (8 * a.remap);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((strips<WAVES, DEPTH, NT_LD, NT_ST, SYNC>), dim3(blocks), dim3(64 * WAVES), 0, 0, a);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((strips<WAVES, DEPTH, NT_LD, NT_ST, SYNC>), dim3(blocks), dim3(64 * WAVES), 0, 0, a);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    CHECK(hipGetLastError());
    return ms / reps;
}

__global__ void __launch_bounds__(256) flat_read2(const float4* __restrict__ s, const float4* __restrict__ r, float* __restrict__ o, size_t n) {
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float4 a = s[i], b = r[i];
        acc += a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w;
    }
    if (acc == 12345.f) o[0] = acc;
}
__global__ void __launch_bounds__(256) flat_copy1(const float4* __restrict__ s, float4* __restrict__ o, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) o[i] = s[i];
}
__global__ void __launch_bounds__(256) flat_write(float4* __restrict__ o, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) o[i] = make_float4(1.f, 2.f, 3.f, 4.f);
}

int main(int argc, char** argv) {
    const int H = 16384, W = 16384, B = 4;
    const long long max_stride = W + 1024;
    const size_t bytes = (size_t)max_stride * H * B * 4;
    float *s, *r, *o;
    CHECK(hipMalloc(&s, bytes)); CHECK(hipMalloc(&r, bytes)); CHECK(hipMalloc(&o, bytes));
    CHECK(hipMemset(s, 0, bytes)); CHECK(hipMemset(r, 0, bytes));
    const double algo = 12.0 * H * W * B;
    {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        const size_t n = (size_t)W * H * B / 4;
        auto timeit = [&](const char* tag, double bytes_moved, auto&& launch) {
            launch(); hipDeviceSynchronize();
            hipEventRecord(e0);
            for (int i = 0; i < 10; ++i) launch();
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 10;
            printf("%-56s %7.3f ms  %6.0f GB/s\n", tag, ms, bytes_moved / ms / 1e6);
        };
        for (int grid : {4096, 65536}) {
            char tag[128];
            snprintf(tag, sizeof tag, "flat 2-read 1-write float4 stream, grid %d", grid);
            timeit(tag, algo, [&] { hipLaunchKernelGGL(flat_copy, dim3(grid), dim3(256), 0, 0, (const float4*)s, (const float4*)r, (float4*)o, n); });
            snprintf(tag, sizeof tag, "flat 2-read 0-write, grid %d", grid);
            timeit(tag, algo * 8 / 12, [&] { hipLaunchKernelGGL(flat_read2, dim3(grid), dim3(256), 0, 0, (const float4*)s, (const float4*)r, o, n); });
            snprintf(tag, sizeof tag, "flat 1-read 1-write copy, grid %d", grid);
            timeit(tag, algo * 8 / 12, [&] { hipLaunchKernelGGL(flat_copy1, dim3(grid), dim3(256), 0, 0, (const float4*)s, (float4*)o, n); });
            snprintf(tag, sizeof tag, "flat 0-read 1-write fill, grid %d", grid);
            timeit(tag, algo * 4 / 12, [&] { hipLaunchKernelGGL(flat_write, dim3(grid), dim3(256), 0, 0, (float4*)o, n); });
        }
    }
    Args a{};
    a.src = s; a.ref = r; a.out = o; a.height = H; a.width = W; a.n_bands = B;
    a.rh = 2; a.n_strips = (W + 247) / 248;
    printf("%-64s %9s %9s   (12 B per pixel*band algorithmic)\n", "strip march", "ms", "GB/s");
    for (long long stride : {(long long)W, (long long)W + 64, (long long)W + 256 + 64}) {
        a.stride = stride; a.band_stride = stride * H;
        for (int seg : {32, 64, 128, 256}) {
            a.seg = seg; a.n_segs = (H + seg - 1) / seg;
            for (int remap : {16, 64}) {
                a.remap = remap;
                auto line = [&](const char* tag, float ms) {
                    printf("%-36s stride %5lld seg %4d remap %3d %9.3f %9.0f\n", tag, stride, seg, remap, ms, algo / ms / 1e6);
                    fflush(stdout);
                };
                line("1 wave/WG depth 1", run<1, 1, false, false>(a, 10));
                line("1 wave/WG depth 2 nt stores", run<1, 2, false, true>(a, 10));
                line("4 waves/WG depth 2 nt stores", run<4, 2, false, true>(a, 10));
                if (stride == W && remap == 16) {
                    line("4 waves/WG lock-step 1 row", run<4, 2, false, true, 1>(a, 10));
                    line("4 waves/WG lock-step 4 rows", run<4, 2, false, true, 4>(a, 10));
                    line("8 waves/WG lock-step 4 rows", run<8, 2, false, true, 4>(a, 10));
                    line("16 waves/WG lock-step 4 rows", run<16, 2, false, true, 4>(a, 10));
                    line("16 waves/WG free", run<16, 2, false, true>(a, 10));
                }
            }
        }
    }
    return 0;
}
