// Micro-benchmark: package power and shader clock the MI355X holds under the ingredients of the fused fit kernel --
// HBM streams, float64 / convert / DPP / packed-float32 instruction loops, LDS traffic -- alone and combined.
// The fused 5x5 gain-offset kernel runs AT the 1400 W package cap with sclk throttled to ~2.0 GHz (tools/power_probe.sh),
// so its time is energy / 1400 W: this tool prices the energy of each ingredient.
// Each phase repeats its launch for `secs` seconds and prints wall-clock start / end (epoch seconds) + work done;
// tools/power_phases.sh samples rocm-smi beside it and joins the two.
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench_power.hip -o tools/ubench_power
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));

static double now() {
    return std::chrono::duration<double>(std::chrono::system_clock::now().time_since_epoch()).count();
}

// ---- memory streams (persistent grid, 4 x 16-byte loads in flight per lane, non-temporal) ---------------------------
__global__ void __launch_bounds__(256) copy21(const f4* __restrict__ s, const f4* __restrict__ r, f4* __restrict__ o, size_t n) {
    const size_t chunk = 4 * 256;
    for (size_t base = (size_t)blockIdx.x * chunk; base < n; base += (size_t)gridDim.x * chunk) {
        f4 a[4], b[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const size_t i = base + (size_t)u * 256 + threadIdx.x;
            a[u] = __builtin_nontemporal_load(s + i);
            b[u] = __builtin_nontemporal_load(r + i);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) __builtin_nontemporal_store(a[u] + b[u], o + base + (size_t)u * 256 + threadIdx.x);
    }
}
__global__ void __launch_bounds__(256) read2(const f4* __restrict__ s, const f4* __restrict__ r, float* __restrict__ o, size_t n) {
    const size_t chunk = 8 * 256;
    f4 acc = {0, 0, 0, 0};
    for (size_t base = (size_t)blockIdx.x * chunk; base < n; base += (size_t)gridDim.x * chunk) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const size_t i = base + (size_t)u * 256 + threadIdx.x;
            acc += __builtin_nontemporal_load(s + i) * __builtin_nontemporal_load(r + i);
        }
    }
    if (acc.x + acc.y + acc.z + acc.w == 123.456f) o[threadIdx.x] = acc.x;
}

// ---- instruction loops: 16 independent chains per lane, ITER x 16 instructions per wave ------------------------------
constexpr int UNROLL = 16;
template <int OP>
__global__ void __launch_bounds__(256) vloop(double* out, int iters, double seed) {
    extern __shared__ char lds[];
    double a[UNROLL];
    float f[UNROLL];
    int n[UNROLL];
    f2 p[UNROLL];
#pragma unroll
    for (int i = 0; i < UNROLL; ++i) {
        // pseudo-random mantissas: the energy of an operation depends on how many bits toggle
        const unsigned long long h = (unsigned long long)(threadIdx.x * 977 + i * 131 + 7) * 0x9e3779b97f4a7c15ull;
        a[i] = 1.0 + (double)(h >> 12) * 0x1p-52 + seed;
        f[i] = 1.0f + (float)(h >> 41) * 0x1p-23f;
        n[i] = (int)(h >> 32);
        p[i] = f2{f[i], f[i] * 1.3f};
    }
    const double c = 1.0 + seed * 0.5 + 0x1.23456789abcdp-3;
    const float cf = 1.0f + 0x1.234568p-3f;
    const f2 cp = {cf, cf * 1.1f};
    const unsigned lbase = threadIdx.x * 16;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < UNROLL; ++i) {
            if constexpr (OP == 0) asm volatile("s_nop 3");
            if constexpr (OP == 1) asm volatile("v_add_f64 %0, %0, %1" : "+v"(a[i]) : "v"(c));
            if constexpr (OP == 2) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(a[i]) : "v"(c));
            if constexpr (OP == 3) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(a[i]) : "v"(f[i]));
            if constexpr (OP == 4) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(f[i]) : "v"(a[i]));
            if constexpr (OP == 5) asm volatile("v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(n[i]) : "v"(n[(i + 1) % UNROLL]));
            if constexpr (OP == 6) asm volatile("v_add_f32 %0, %0, %1" : "+v"(f[i]) : "v"(cf));
            if constexpr (OP == 7) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p[i]) : "v"(cp));
            if constexpr (OP == 8) asm volatile("v_rcp_f64 %0, %0" : "+v"(a[i]));
            if constexpr (OP == 9) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a[i]) : "v"(c));
            if constexpr (OP == 10) {  // LDS: one 16-byte write + one 16-byte read per lane
                asm volatile("ds_write_b128 %0, %1" :: "v"(lbase), "v"(*reinterpret_cast<f4*>(&a[i & ~1])) : "memory");
                f4 t;
                asm volatile("ds_read_b128 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(t) : "v"(lbase) : "memory");
                f[i] = t.x;
            }
            if constexpr (OP == 11) {  // LDS reads only
                f4 t;
                asm volatile("ds_read_b128 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(t) : "v"(lbase) : "memory");
                f[i] = t.x;
            }
            if constexpr (OP == 12) asm volatile("v_mov_b32 %0, %1" : "=v"(n[i]) : "v"(n[(i + 1) % UNROLL]));
            if constexpr (OP == 13) asm volatile("v_cmp_lt_f32 vcc, %0, %1" :: "v"(f[i]), "v"(cf) : "vcc");
        }
    }
    double sum = 0.0;
#pragma unroll
    for (int i = 0; i < UNROLL; ++i) sum += a[i] + f[i] + n[i] + p[i].x;
    if (sum == 0.12345) out[threadIdx.x] = sum;
}

static const char* OPNAME[] = {"s_nop 3 (clocked, nothing issued to the VALU)", "v_add_f64", "v_fma_f64", "v_cvt_f64_f32", "v_cvt_f32_f64",
                               "v_mov_b32_dpp wave_shr", "v_add_f32", "v_pk_fma_f32", "v_rcp_f64", "v_mul_f64",
                               "ds_write_b128 + ds_read_b128", "ds_read_b128", "v_mov_b32", "v_cmp_lt_f32"};

template <int OP>
static void launch_v(double* out, int grid, int iters, hipStream_t st) {
    hipLaunchKernelGGL((vloop<OP>), dim3(grid), dim3(256), 4096, st, out, iters, 0.0);
}
typedef void (*launch_fn)(double*, int, int, hipStream_t);
static launch_fn LAUNCH[] = {launch_v<0>, launch_v<1>, launch_v<2>, launch_v<3>, launch_v<4>, launch_v<5>, launch_v<6>,
                             launch_v<7>, launch_v<8>, launch_v<9>, launch_v<10>, launch_v<11>, launch_v<12>, launch_v<13>};
constexpr int N_OPS = sizeof(LAUNCH) / sizeof(LAUNCH[0]);

int main(int argc, char** argv) {
    const double secs = argc > 1 ? atof(argv[1]) : 4.0;
    const size_t px = (size_t)16384 * 16384 * 4;
    const size_t n = px / 4, bytes = px * 4;
    char *s, *r, *o;
    double* out;
    CHECK(hipMalloc(&s, bytes)); CHECK(hipMalloc(&r, bytes)); CHECK(hipMalloc(&o, bytes)); CHECK(hipMalloc(&out, 4096));
    // random-looking payload (bit toggling on the wires is part of the energy)
    {
        const size_t words = bytes / 4;
        unsigned* h = (unsigned*)malloc(64 << 20);
        unsigned long long z = 88172645463325252ull;
        for (size_t i = 0; i < (64u << 20) / 4; ++i) { z ^= z << 13; z ^= z >> 7; z ^= z << 17; h[i] = 0x3f000000u | ((unsigned)z & 0x007fffffu); }
        for (size_t off = 0; off < bytes; off += 64u << 20) {
            CHECK(hipMemcpy(s + off, h, 64u << 20, hipMemcpyHostToDevice));
            CHECK(hipMemcpy(r + off, h, 64u << 20, hipMemcpyHostToDevice));
        }
        (void)words;
        free(h);
    }
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    hipStream_t st, st2;
    CHECK(hipStreamCreate(&st)); CHECK(hipStreamCreate(&st2));
    printf("# device %s, %d CUs, %.1f s per phase\n", prop.gcnArchName, cus, secs);
    auto phase = [&](const char* name, double unit_per_launch, const char* unit, auto&& launch) {
        launch(); hipDeviceSynchronize();
        const double t0 = now();
        long launches = 0;
        while (now() - t0 < secs) {
            for (int k = 0; k < 8; ++k) launch();
            launches += 8;
            hipDeviceSynchronize();
        }
        const double t1 = now();
        printf("PHASE\t%s\t%.3f\t%.3f\t%ld\t%.6g\t%s\n", name, t0, t1, launches, unit_per_launch * launches / (t1 - t0), unit);
        fflush(stdout);
    };
    // idle gap
    { const double t0 = now(); while (now() - t0 < secs) { } printf("PHASE\tidle\t%.3f\t%.3f\t0\t0\t-\n", t0, now()); fflush(stdout); }
    phase("stream 2-read 1-write nt (GB/s)", 3.0 * bytes * 1e-9, "GB/s", [&] { hipLaunchKernelGGL(copy21, dim3(cus * 4), dim3(256), 0, st, (const f4*)s, (const f4*)r, (f4*)o, n); });
    phase("stream 2-read 0-write nt (GB/s)", 2.0 * bytes * 1e-9, "GB/s", [&] { hipLaunchKernelGGL(read2, dim3(cus * 4), dim3(256), 0, st, (const f4*)s, (const f4*)r, (float*)o, n); });
    const int iters = 4000;
    const int grid = cus * 8;  // 8 workgroups of 4 waves per CU = 8 waves per SIMD
    const double ginstr = (double)grid * 4 * iters * UNROLL * 1e-9;  // wave-instructions per launch, in G
    for (int op = 0; op < N_OPS; ++op) {
        char name[128];
        snprintf(name, sizeof name, "valu %s (G wave-instr/s)", OPNAME[op]);
        phase(name, ginstr, "Ginstr/s", [&] { LAUNCH[op](out, grid, iters, st); });
    }
    // combined: the stream on one queue, a float64 loop on 4 waves per SIMD on another (both share every CU)
    for (int op : {1, 2, 3}) {
        char name[128];
        snprintf(name, sizeof name, "stream 2R1W + valu %s on 4 waves/SIMD (GB/s of the stream)", OPNAME[op]);
        phase(name, 3.0 * bytes * 1e-9, "GB/s", [&] {
            LAUNCH[op](out, cus * 4, 16000, st2);
            hipLaunchKernelGGL(copy21, dim3(cus * 4), dim3(256), 0, st, (const f4*)s, (const f4*)r, (f4*)o, n);
        });
    }
    hipFree(s), hipFree(r), hipFree(o), hipFree(out);
    return 0;
}
