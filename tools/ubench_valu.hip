// Micro-benchmark: sustained VALU issue cost (cycles per wave-instruction per SIMD) of the instruction kinds the
// fit kernel is made of, at 1/2/4 waves per SIMD, the clock the chip holds meanwhile, and the accuracy of v_rcp_f64.
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench_valu.hip -o tools/ubench_valu
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

constexpr int ITER = 20000;
constexpr int UNROLL = 16;  // independent chains per lane
constexpr int N_OPS = 50;

typedef float f2 __attribute__((ext_vector_type(2)));
// which per-lane arrays an op keeps live (a = 1, f = 2, n = 4, p = 8, u = 16): the others fold to constants, so every
// kernel stays under 64 VGPRs and 8 waves fit on a SIMD
constexpr int live_mask(int op) {
    switch (op) {
        case 0: case 1: case 2: case 5: case 6: case 7: case 8: case 21: case 25: case 36: case 40: case 45: case 48: return 1;
        case 3: case 4: case 43: return 1 | 2;
        case 33: case 39: case 44: case 49: return 1 | 4;
        case 9: case 10: case 13: case 16: case 17: case 18: case 19: case 20: case 22: case 31: case 34: return 2;
        case 23: return 2 | 4;
        case 26: case 27: case 28: case 47: return 8;
        case 29: return 16;
        case 37: return 16 | 4;
        default: return 4;
    }
}

template <int OP>
__global__ void __launch_bounds__(256) k(double* out, unsigned long long* clk, int iters, double seed) {
    extern __shared__ char lds_pad[];  // sized by the host so that exactly `waves_per_simd` workgroups fit on a CU
    if (seed == -1.0) out[1] = lds_pad[threadIdx.x];
    double a[UNROLL];
    float f[UNROLL];
    int n[UNROLL];
    f2 p[UNROLL];
    unsigned long long u[UNROLL];
    constexpr int LM = live_mask(OP);
#pragma unroll
    for (int i = 0; i < UNROLL; ++i) {
        const double v = seed + i + threadIdx.x;
        a[i] = (LM & 1) ? v : 1.0; f[i] = (LM & 2) ? (float)v : 1.f; n[i] = (LM & 4) ? (int)v : 1;
        p[i] = (LM & 8) ? f2{(float)v, (float)v + 1.f} : f2{1.f, 1.f};
        u[i] = (LM & 16) ? (unsigned long long)v * 77ull : 1ull;
    }
    const double c = seed * 0.5 + 1.0;
    const float cf = (float)c;
    const f2 cp = {cf, cf + 0.25f};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < UNROLL; ++i) {
            if constexpr (OP == 0) asm volatile("v_add_f64 %0, %0, %1" : "+v"(a[i]) : "v"(c));
            if constexpr (OP == 1) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(a[i]) : "v"(c));
            if constexpr (OP == 2) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a[i]) : "v"(c));
            if constexpr (OP == 3) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(a[i]) : "v"(f[i]));
            if constexpr (OP == 4) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(f[i]) : "v"(a[i]));
            if constexpr (OP == 5) asm volatile("v_rcp_f64 %0, %0" : "+v"(a[i]));
            if constexpr (OP == 6) asm volatile("v_div_scale_f64 %0, vcc, %0, %1, %0" : "+v"(a[i]) : "v"(c) : "vcc");
            if constexpr (OP == 7) asm volatile("v_div_fmas_f64 %0, %0, %1, %1" : "+v"(a[i]) : "v"(c) : "vcc");
            if constexpr (OP == 8) asm volatile("v_div_fixup_f64 %0, %0, %1, %1" : "+v"(a[i]) : "v"(c));
            if constexpr (OP == 9) asm volatile("v_add_f32 %0, %0, %1" : "+v"(f[i]) : "v"(cf));
            if constexpr (OP == 10) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(f[i]) : "v"(cf));
            if constexpr (OP == 11) asm volatile("v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(n[i]) : "v"(n[(i + 1) % UNROLL]));
            if constexpr (OP == 12) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(n[i]) : "v"(n[(i + 1) % UNROLL]) : "vcc");
            if constexpr (OP == 13) asm volatile("v_rcp_f32 %0, %0" : "+v"(f[i]));
            if constexpr (OP == 14) asm volatile("v_mov_b32 %0, %1" : "=v"(n[i]) : "v"(n[(i + 1) % UNROLL]));
            if constexpr (OP == 15) asm volatile("v_add_u32 %0, %0, %1" : "+v"(n[i]) : "v"(n[(i + 1) % UNROLL]));
            if constexpr (OP == 16) asm volatile("v_cmp_lt_f32 vcc, %0, %1" :: "v"(f[i]), "v"(cf) : "vcc");
            if constexpr (OP == 17) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(f[i]) : "v"(cf));
            if constexpr (OP == 18) asm volatile("v_div_scale_f32 %0, vcc, %0, %1, %0" : "+v"(f[i]) : "v"(cf) : "vcc");
            if constexpr (OP == 19) asm volatile("v_div_fmas_f32 %0, %0, %1, %1" : "+v"(f[i]) : "v"(cf) : "vcc");
            if constexpr (OP == 20) asm volatile("v_div_fixup_f32 %0, %0, %1, %1" : "+v"(f[i]) : "v"(cf));
            if constexpr (OP == 21) a[i] = a[i] / c;                 // full IEEE f64 division (compiler expansion)
            if constexpr (OP == 22) f[i] = f[i] / cf;                // full IEEE f32 division
            if constexpr (OP == 23) asm volatile("v_cvt_f32_ubyte0 %0, %1" : "=v"(f[i]) : "v"(n[i]));
            if constexpr (OP == 24) asm volatile("v_bfe_u32 %0, %1, 8, 8" : "=v"(n[i]) : "v"(n[(i + 1) % UNROLL]));
            if constexpr (OP == 25) asm volatile("v_cmp_lt_f64 vcc, %0, %1" :: "v"(a[i]), "v"(c) : "vcc");
            if constexpr (OP == 26) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(cp));
            if constexpr (OP == 27) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(cp));
            if constexpr (OP == 28) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p[i]) : "v"(cp));
            if constexpr (OP == 29) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(u[i]) : "v"(u[(i + 1) % UNROLL]));
            if constexpr (OP == 30) asm volatile("v_add_co_u32 %0, vcc, %0, %1" : "+v"(n[i]) : "v"(n[(i + 1) % UNROLL]) : "vcc");
            if constexpr (OP == 31) asm volatile("v_add_f32_dpp %0, %1, %0 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(f[i]) : "v"(f[(i + 1) % UNROLL]));
            if constexpr (OP == 32) asm volatile("v_add_u32_dpp %0, %1, %0 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(n[i]) : "v"(n[(i + 1) % UNROLL]));
            if constexpr (OP == 33) asm volatile("v_cvt_f64_i32 %0, %1" : "=v"(a[i]) : "v"(n[i]));
            if constexpr (OP == 34) asm volatile("v_min3_f32 %0, %0, %1, %2" : "+v"(f[i]) : "v"(f[(i + 1) % UNROLL]), "v"(cf));
            if constexpr (OP == 35) asm volatile("v_and_b32 %0, %0, %1" : "+v"(n[i]) : "v"(n[(i + 1) % UNROLL]));
            if constexpr (OP == 36) asm volatile("v_mov_b64 %0, %1" : "=v"(a[i]) : "v"(a[(i + 1) % UNROLL]));
            if constexpr (OP == 37) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(u[i]) : "v"(n[i]), "v"(n[(i + 1) % UNROLL]) : "vcc");
            if constexpr (OP == 38) asm volatile("v_cmp_lt_u32 vcc, %0, %1" :: "v"(n[i]), "v"(n[(i + 1) % UNROLL]) : "vcc");
            if constexpr (OP == 39) asm volatile("v_ldexp_f64 %0, %0, %1" : "+v"(a[i]) : "v"(n[i]));
            if constexpr (OP == 40) asm volatile("v_max_f64 %0, %0, %1" : "+v"(a[i]) : "v"(c));
            if constexpr (OP == 41) asm volatile("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(n[i]) : "v"(n[(i + 1) % UNROLL]));
            if constexpr (OP == 42) asm volatile("v_add_co_u32_dpp %0, vcc, %1, %0 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(n[i]) : "v"(n[(i + 1) % UNROLL]) : "vcc");
            if constexpr (OP == 43) {  // 1:1 mix of f64 add and f32 add: does the f32 op hide behind the f64 one?
                asm volatile("v_add_f64 %0, %0, %1" : "+v"(a[i]) : "v"(c));
                asm volatile("v_add_f32 %0, %0, %1" : "+v"(f[i]) : "v"(cf));
            }
            if constexpr (OP == 44) {  // f64 add + DPP move
                asm volatile("v_add_f64 %0, %0, %1" : "+v"(a[i]) : "v"(c));
                asm volatile("v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(n[i]) : "v"(n[(i + 1) % UNROLL]));
            }
            if constexpr (OP == 45) asm volatile("v_rsq_f64 %0, %0" : "+v"(a[i]));
            if constexpr (OP == 46) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(n[i]), "+v"(n[(i + 1) % UNROLL]));
            if constexpr (OP == 47) asm volatile("v_pk_add_f32 %0, %0, %1 op_sel_hi:[1,0]" : "+v"(p[i]) : "v"(cp));
            if constexpr (OP == 48) asm volatile("v_fmac_f64 %0, %1, %1" : "+v"(a[i]) : "v"(c));
            if constexpr (OP == 49) asm volatile("v_cvt_f64_u32 %0, %1" : "=v"(a[i]) : "v"(n[i]));
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    double s = 0;
#pragma unroll
    for (int i = 0; i < UNROLL; ++i) {
        if (LM & 1) s += a[i];
        if (LM & 2) s += f[i];
        if (LM & 4) s += n[i];
        if (LM & 8) s += p[i].x + p[i].y;
        if (LM & 16) s += (double)u[i];
    }
    if (s == 12345.678) out[0] = s;
    if ((threadIdx.x & 63) == 0) { atomicAdd(clk + 0, t1 - t0); atomicAdd(clk + 1, r1 - r0); atomicAdd(clk + 2, 1ull); }
}

struct Res { float ns; double cyc; double ghz; };

template <int OP>
Res run(double* d, unsigned long long* clk, int waves_per_simd, int n_cu) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    // 256-thread workgroups = one wave per SIMD; the LDS request admits exactly `waves_per_simd` of them per CU
    const int blocks = n_cu * waves_per_simd;
    const size_t lds = (size_t)160 * 1024 / waves_per_simd - 512;
    hipFuncSetAttribute(reinterpret_cast<const void*>(&k<OP>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), lds, 0, d, clk, 10, 1.5);
    hipDeviceSynchronize();
    hipMemset(clk, 0, 64);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), lds, 0, d, clk, ITER, 1.5);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[3];
    hipMemcpy(h, clk, sizeof(h), hipMemcpyDeviceToHost);
    const int per = (OP == 43 || OP == 44) ? 2 : 1;
    Res r;
    r.ns = ms * 1e6f / ((float)ITER * UNROLL * per * waves_per_simd);
    r.cyc = (double)h[0] / (double)h[2] / ((double)ITER * UNROLL * per * waves_per_simd);  // shader cycles of one wave per wave-instruction issued on its SIMD
    r.ghz = h[1] ? (double)h[0] / (double)h[1] * 0.1 : 0.0;                  // s_memrealtime ticks at 100 MHz
    return r;
}

// accuracy of v_rcp_f64 and of one / two Newton steps on it
__global__ void rcp_acc(const double* x, double* r0, double* r1, double* r2, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double d = x[i];
    double y;
    asm volatile("v_rcp_f64 %0, %1" : "=v"(y) : "v"(d));
    r0[i] = y;
    double e = __fma_rn(-d, y, 1.0);
    y = __fma_rn(y, e, y);
    r1[i] = y;
    e = __fma_rn(-d, y, 1.0);
    y = __fma_rn(y, e, y);
    r2[i] = y;
}

int main(int argc, char** argv) {
    double* d; CHECK(hipMalloc(&d, 64));
    unsigned long long* clk; CHECK(hipMalloc(&clk, 64));
    hipDeviceProp_t p; CHECK(hipGetDeviceProperties(&p, 0));
    const int n_cu = p.multiProcessorCount;
    printf("device %s, %d CUs, clock %d kHz\n", p.gcnArchName, n_cu, p.clockRate);
    const char* names[N_OPS] = {"v_add_f64","v_fma_f64","v_mul_f64","v_cvt_f64_f32","v_cvt_f32_f64","v_rcp_f64","v_div_scale_f64","v_div_fmas_f64","v_div_fixup_f64","v_add_f32","v_fma_f32","v_mov_b32_dpp wave_shr","v_cndmask_b32","v_rcp_f32","v_mov_b32","v_add_u32","v_cmp_lt_f32","v_mul_f32","v_div_scale_f32","v_div_fmas_f32","v_div_fixup_f32","f64 div (full)","f32 div (full)","v_cvt_f32_ubyte0","v_bfe_u32","v_cmp_lt_f64",
        "v_pk_add_f32","v_pk_mul_f32","v_pk_fma_f32","v_lshl_add_u64","v_add_co_u32","v_add_f32_dpp wave_shr","v_add_u32_dpp wave_shr","v_cvt_f64_i32","v_min3_f32","v_and_b32","v_mov_b64","v_mad_u64_u32","v_cmp_lt_u32","v_ldexp_f64","v_max_f64","v_mov_b32_dpp row_shr","v_add_co_u32_dpp","mix add_f64+add_f32","mix add_f64+dpp","v_rsq_f64","v_permlane32_swap","v_pk_add_f32 op_sel","v_fmac_f64","v_cvt_f64_u32"};
    printf("%-24s %7s %7s %7s %7s | %7s %7s %7s %7s | %6s   (wall ns and in-kernel shader cycles per wave-instruction per SIMD at 1/2/4/8 waves per SIMD; GHz at 8)\n", "op", "ns@1", "ns@2", "ns@4", "ns@8", "cyc@1", "cyc@2", "cyc@4", "cyc@8", "GHz");
    for (int op = 0; op < N_OPS; ++op) {
        Res r[4]; int ws[4] = {1, 2, 4, 8};
        for (int j = 0; j < 4; ++j) {
            switch (op) {
#define C(N) case N: r[j] = run<N>(d, clk, ws[j], n_cu); break;
                C(0) C(1) C(2) C(3) C(4) C(5) C(6) C(7) C(8) C(9) C(10) C(11) C(12) C(13) C(14) C(15) C(16) C(17) C(18) C(19) C(20) C(21) C(22) C(23) C(24) C(25)
                C(26) C(27) C(28) C(29) C(30) C(31) C(32) C(33) C(34) C(35) C(36) C(37) C(38) C(39) C(40) C(41) C(42) C(43) C(44) C(45) C(46) C(47) C(48) C(49)
            }
        }
        printf("%-24s %7.3f %7.3f %7.3f %7.3f | %7.2f %7.2f %7.2f %7.2f | %6.3f\n", names[op], r[0].ns, r[1].ns, r[2].ns, r[3].ns, r[0].cyc, r[1].cyc, r[2].cyc, r[3].cyc, r[3].ghz);
        fflush(stdout);
    }
    {   // v_rcp_f64 accuracy
        const int n = 1 << 22;
        std::vector<double> x(n), y0(n), y1(n), y2(n);
        unsigned long long s = 0x9e3779b97f4a7c15ull;
        for (int i = 0; i < n; ++i) {
            s ^= s << 13; s ^= s >> 7; s ^= s << 17;
            const double m = 1.0 + (double)(s >> 11) * 0x1p-53;
            x[i] = ldexp(m, (int)(s % 41) - 20);
        }
        double *dx, *d0, *d1, *d2;
        CHECK(hipMalloc(&dx, n * 8)); CHECK(hipMalloc(&d0, n * 8)); CHECK(hipMalloc(&d1, n * 8)); CHECK(hipMalloc(&d2, n * 8));
        CHECK(hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(rcp_acc, dim3(n / 256), dim3(256), 0, 0, dx, d0, d1, d2, n);
        CHECK(hipMemcpy(y0.data(), d0, n * 8, hipMemcpyDeviceToHost));
        CHECK(hipMemcpy(y1.data(), d1, n * 8, hipMemcpyDeviceToHost));
        CHECK(hipMemcpy(y2.data(), d2, n * 8, hipMemcpyDeviceToHost));
        long double m0 = 0, m1 = 0, m2 = 0;
        for (int i = 0; i < n; ++i) {
            const long double t = 1.0L / (long double)x[i];
            m0 = fmaxl(m0, fabsl(((long double)y0[i] - t) / t));
            m1 = fmaxl(m1, fabsl(((long double)y1[i] - t) / t));
            m2 = fmaxl(m2, fabsl(((long double)y2[i] - t) / t));
        }
        printf("v_rcp_f64 max relative error: raw 2^%.2f, after one Newton step 2^%.2f, after two 2^%.2f\n",
               (double)log2l(m0), (double)log2l(m1), (double)log2l(m2));
    }
    return 0;
}
