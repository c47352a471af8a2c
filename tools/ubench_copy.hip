// Micro-benchmark: what a FLAT stream of the fit kernel's bytes reaches on this MI355X -- the line the strip-march
// pattern (tools/ubench_strips.hip) and the fused kernel are compared with.  /opt/skills/guides/MI355X_MICROARCH.md
// records 6.29 TB/s for a float4 copy (bytes read + bytes written); round 2's flat kernels (one 16-byte access in
// flight per lane per array) stopped at 4.7-5.4.  Here: persistent grids of CUs x k workgroups, U independent 16-byte
// loads in flight per lane before the first store, default / non-temporal policies, 2 MB-aligned planes.
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench_copy.hip -o tools/ubench_copy
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

typedef float f4 __attribute__((ext_vector_type(4)));

template <bool NT> __device__ __forceinline__ f4 ld(const f4* p) {
    if constexpr (NT) return __builtin_nontemporal_load(p);
    else return *p;
}
template <bool NT> __device__ __forceinline__ void st(f4* p, f4 v) {
    if constexpr (NT) __builtin_nontemporal_store(v, p);
    else *p = v;
}

// 1 read + 1 write.  Each workgroup walks chunks of U * 256 quads; all U loads of a lane are issued before its stores.
template <int U, bool NTL, bool NTS>
__global__ void __launch_bounds__(256) copy11(const f4* __restrict__ s, f4* __restrict__ o, size_t n) {
    const size_t chunk = (size_t)U * 256;
    for (size_t base = (size_t)blockIdx.x * chunk; base < n; base += (size_t)gridDim.x * chunk) {
        f4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const size_t i = base + (size_t)u * 256 + threadIdx.x;
            v[u] = i < n ? ld<NTL>(s + i) : f4{0, 0, 0, 0};
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const size_t i = base + (size_t)u * 256 + threadIdx.x;
            if (i < n) st<NTS>(o + i, v[u]);
        }
    }
}

// 2 reads + 1 write (the fused kernel's 12 bytes per pixel): o = s + r
template <int U, bool NTL, bool NTS>
__global__ void __launch_bounds__(256) copy21(const f4* __restrict__ s, const f4* __restrict__ r, f4* __restrict__ o, size_t n) {
    const size_t chunk = (size_t)U * 256;
    for (size_t base = (size_t)blockIdx.x * chunk; base < n; base += (size_t)gridDim.x * chunk) {
        f4 a[U], b[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const size_t i = base + (size_t)u * 256 + threadIdx.x;
            a[u] = i < n ? ld<NTL>(s + i) : f4{0, 0, 0, 0};
            b[u] = i < n ? ld<NTL>(r + i) : f4{0, 0, 0, 0};
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const size_t i = base + (size_t)u * 256 + threadIdx.x;
            if (i < n) st<NTS>(o + i, a[u] + b[u]);
        }
    }
}

// 2 reads, result folded into one float per lane (no stores to speak of)
template <int U, bool NTL>
__global__ void __launch_bounds__(256) read2(const f4* __restrict__ s, const f4* __restrict__ r, float* __restrict__ o, size_t n) {
    const size_t chunk = (size_t)U * 256;
    f4 acc = {0, 0, 0, 0};
    for (size_t base = (size_t)blockIdx.x * chunk; base < n; base += (size_t)gridDim.x * chunk) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const size_t i = base + (size_t)u * 256 + threadIdx.x;
            if (i < n) acc += ld<NTL>(s + i) * ld<NTL>(r + i);
        }
    }
    if (acc.x + acc.y + acc.z + acc.w == 123.456f) o[threadIdx.x] = acc.x;
}

template <int U, bool NTS>
__global__ void __launch_bounds__(256) fill(f4* __restrict__ o, size_t n) {
    const size_t chunk = (size_t)U * 256;
    for (size_t base = (size_t)blockIdx.x * chunk; base < n; base += (size_t)gridDim.x * chunk) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const size_t i = base + (size_t)u * 256 + threadIdx.x;
            if (i < n) st<NTS>(o + i, f4{1.f, 2.f, 3.f, 4.f});
        }
    }
}

template <typename F>
static double timeit(const char* tag, double bytes, F&& launch, int reps = 20) {
    for (int i = 0; i < 3; ++i) launch();
    hipDeviceSynchronize();
    std::vector<float> ms(reps);
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    for (int i = 0; i < reps; ++i) {
        hipEventRecord(e0, 0);
        launch();
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms[i], e0, e1);
    }
    std::sort(ms.begin(), ms.end());
    const double med = ms[reps / 2], best = ms[0];
    printf("%-66s median %7.3f ms %6.0f GB/s   best %7.3f ms %6.0f GB/s\n", tag, med, bytes / med * 1e-6, best, bytes / best * 1e-6);
    fflush(stdout);
    hipEventDestroy(e0), hipEventDestroy(e1);
    return bytes / med * 1e-6;
}

int main(int argc, char** argv) {
    const size_t px = (size_t)16384 * 16384 * 4;  // the headline launch: 4 bands of 16384^2 float32
    const size_t n = px / 4;                      // quads
    const size_t bytes = px * 4;
    char *s, *r, *o;
    // hipMalloc of GB-sized buffers is 2 MB-aligned; assert it
    CHECK(hipMalloc(&s, bytes)); CHECK(hipMalloc(&r, bytes)); CHECK(hipMalloc(&o, bytes));
    printf("plane origins mod 2 MB: %zu %zu %zu\n", (size_t)s % (2u << 20), (size_t)r % (2u << 20), (size_t)o % (2u << 20));
    CHECK(hipMemset(s, 1, bytes)); CHECK(hipMemset(r, 2, bytes)); CHECK(hipMemset(o, 0, bytes));
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    printf("device %s, %d CUs; %.3f GB per plane\n", prop.gcnArchName, cus, bytes * 1e-9);
    char tag[160];
    timeit("hipMemcpyDtoD (runtime's own copy), bytes read + written", 2.0 * bytes, [&] { (void)hipMemcpyAsync(o, s, bytes, hipMemcpyDeviceToDevice, 0); });
#define RUN11(U, NTL, NTS, G) do { snprintf(tag, sizeof tag, "1-read 1-write  U=%d loads=%s stores=%s grid=%dxCUs", U, NTL ? "nt" : "default", NTS ? "nt" : "default", G); \
        timeit(tag, 2.0 * bytes, [&] { hipLaunchKernelGGL((copy11<U, NTL, NTS>), dim3(cus * G), dim3(256), 0, 0, (const f4*)s, (f4*)o, n); }); } while (0)
#define RUN21(U, NTL, NTS, G) do { snprintf(tag, sizeof tag, "2-read 1-write  U=%d loads=%s stores=%s grid=%dxCUs", U, NTL ? "nt" : "default", NTS ? "nt" : "default", G); \
        timeit(tag, 3.0 * bytes, [&] { hipLaunchKernelGGL((copy21<U, NTL, NTS>), dim3(cus * G), dim3(256), 0, 0, (const f4*)s, (const f4*)r, (f4*)o, n); }); } while (0)
#define RUNR2(U, NTL, G) do { snprintf(tag, sizeof tag, "2-read 0-write  U=%d loads=%s grid=%dxCUs", U, NTL ? "nt" : "default", G); \
        timeit(tag, 2.0 * bytes, [&] { hipLaunchKernelGGL((read2<U, NTL>), dim3(cus * G), dim3(256), 0, 0, (const f4*)s, (const f4*)r, (float*)o, n); }); } while (0)
#define RUNF(U, NTS, G) do { snprintf(tag, sizeof tag, "0-read 1-write  U=%d stores=%s grid=%dxCUs", U, NTS ? "nt" : "default", G); \
        timeit(tag, 1.0 * bytes, [&] { hipLaunchKernelGGL((fill<U, NTS>), dim3(cus * G), dim3(256), 0, 0, (f4*)o, n); }); } while (0)
    for (int g : {2, 4, 8, 16}) {
        switch (g) {
#define GCASE(G) case G: \
            RUN11(1, false, false, G); RUN11(4, false, false, G); RUN11(8, false, false, G); RUN11(4, true, true, G); RUN11(8, true, true, G); RUN11(8, false, true, G); \
            RUN21(2, false, false, G); RUN21(4, false, false, G); RUN21(4, true, true, G); RUN21(4, false, true, G); RUN21(8, true, true, G); \
            RUNR2(4, false, G); RUNR2(8, true, G); RUNF(4, false, G); RUNF(8, true, G); break;
            GCASE(2) GCASE(4) GCASE(8) GCASE(16)
        }
    }
    // non-persistent reference: one chunk per workgroup
    {
        const int grid = (int)((n + 4 * 256 - 1) / (4 * 256));
        snprintf(tag, sizeof tag, "1-read 1-write  U=4 default policies, one chunk per workgroup (grid %d)", grid);
        timeit(tag, 2.0 * bytes, [&] { hipLaunchKernelGGL((copy11<4, false, false>), dim3(grid), dim3(256), 0, 0, (const f4*)s, (f4*)o, n); });
        snprintf(tag, sizeof tag, "2-read 1-write  U=4 nt stores, one chunk per workgroup (grid %d)", grid);
        timeit(tag, 3.0 * bytes, [&] { hipLaunchKernelGGL((copy21<4, false, true>), dim3(grid), dim3(256), 0, 0, (const f4*)s, (const f4*)r, (f4*)o, n); });
    }
    hipFree(s), hipFree(r), hipFree(o);
    return 0;
}
