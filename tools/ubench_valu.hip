// Micro-benchmark: sustained VALU issue cost (cycles per wave-instruction per SIMD) of the instruction kinds the
// fit kernel is made of, at 1/2/4 waves per SIMD.  Build: hipcc --offload-arch=gfx950 -O3 tools/ubench_valu.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
#include <string>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

constexpr int ITER = 2000;
constexpr int UNROLL = 16;  // independent chains per lane

template <int OP>
__global__ void __launch_bounds__(64) k(double* out, int iters, double seed) {
    double a[UNROLL];
    float f[UNROLL];
    int n[UNROLL];
#pragma unroll
    for (int i = 0; i < UNROLL; ++i) { a[i] = seed + i + threadIdx.x; f[i] = (float)a[i]; n[i] = (int)a[i]; }
    const double c = seed * 0.5 + 1.0;
    const float cf = (float)c;
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
        }
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < UNROLL; ++i) s += a[i] + f[i] + n[i];
    if (s == 12345.678) out[0] = s;
}

template <int OP>
float run(double* d, int waves_per_simd, int n_cu) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int blocks = n_cu * 4 * waves_per_simd;  // one 64-thread block per wave slot
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(64), 0, 0, d, 10, 1.5);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(64), 0, 0, d, ITER, 1.5);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main() {
    double* d; CHECK(hipMalloc(&d, 64));
    hipDeviceProp_t p; CHECK(hipGetDeviceProperties(&p, 0));
    const int n_cu = p.multiProcessorCount;
    printf("device %s, %d CUs, clock %d kHz\n", p.gcnArchName, n_cu, p.clockRate);
    const char* names[] = {"v_add_f64","v_fma_f64","v_mul_f64","v_cvt_f64_f32","v_cvt_f32_f64","v_rcp_f64","v_div_scale_f64","v_div_fmas_f64","v_div_fixup_f64","v_add_f32","v_fma_f32","v_mov_b32_dpp","v_cndmask_b32","v_rcp_f32","v_mov_b32","v_add_u32","v_cmp_lt_f32","v_mul_f32","v_div_scale_f32","v_div_fmas_f32","v_div_fixup_f32","f64 div (full)","f32 div (full)","v_cvt_f32_ubyte0","v_bfe_u32","v_cmp_lt_f64"};
    printf("%-18s %10s %10s %10s   (ns per wave-instruction per SIMD; x clock GHz = cycles)\n", "op", "1 w/SIMD", "2 w/SIMD", "4 w/SIMD");
    for (int op = 0; op < 26; ++op) {
        float r[3]; int ws[3] = {1, 2, 4};
        for (int j = 0; j < 3; ++j) {
            float ms = 0;
            switch (op) {
#define C(N) case N: ms = run<N>(d, ws[j], n_cu); break;
                C(0) C(1) C(2) C(3) C(4) C(5) C(6) C(7) C(8) C(9) C(10) C(11) C(12) C(13) C(14) C(15) C(16) C(17) C(18) C(19) C(20) C(21) C(22) C(23) C(24) C(25)
            }
            // per SIMD: ws[j] waves each issuing ITER*UNROLL instrs
            r[j] = ms * 1e6f / ((float)ITER * UNROLL * ws[j]);
        }
        printf("%-18s %10.3f %10.3f %10.3f\n", names[op], r[0], r[1], r[2]);
    }
    return 0;
}
