// Micro-benchmark: what a value from another lane costs on gfx950 -- ds_bpermute_b32 (LDS crossbar, shared by the four SIMDs of
// a CU) against v_mov_b32_dpp (VALU, per SIMD), alone and mixed with float64 adds, at 1 / 2 / 3 waves per SIMD.
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench_xlane.hip -o tools/ubench_xlane
#include <hip/hip_runtime.h>
#include <cstdio>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
constexpr int ITER = 4000, U = 8;

// OP 0: U ds_bpermute; 1: U dpp moves; 2: U bpermute + 2U f64 adds; 3: U bpermute + 4U f64 adds; 4: 2U f64 adds; 5: U ds_read_b128; 6: U ds_swizzle
template <int OP>
__global__ void __launch_bounds__(256) k(double* out, unsigned long long* clk, int iters, int seed) {
    extern __shared__ char lds[];
    int n[U], addr = ((threadIdx.x + seed) & 63) << 2;
    double a[2 * U];
    float4 q[U];
#pragma unroll
    for (int i = 0; i < U; ++i) n[i] = threadIdx.x * 3 + i + seed, a[2 * i] = seed + i, a[2 * i + 1] = seed - i, q[i] = make_float4(0, 0, 0, 0);
    const double c = seed * 0.5 + 1.0;
    reinterpret_cast<float4*>(lds)[threadIdx.x] = make_float4(1, 2, 3, 4);
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < U; ++i) {
            if constexpr (OP == 0 || OP == 2 || OP == 3) asm volatile("ds_bpermute_b32 %0, %1, %2" : "=v"(n[i]) : "v"(addr), "v"(n[(i + 1) % U]));
            if constexpr (OP == 1) asm volatile("v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(n[i]) : "v"(n[(i + 1) % U]));
            if constexpr (OP == 2 || OP == 3 || OP == 4) {
                asm volatile("v_add_f64 %0, %0, %1" : "+v"(a[2 * i]) : "v"(c));
                asm volatile("v_add_f64 %0, %0, %1" : "+v"(a[2 * i + 1]) : "v"(c));
            }
            if constexpr (OP == 3) {
                asm volatile("v_add_f64 %0, %0, %1" : "+v"(a[2 * i]) : "v"(c));
                asm volatile("v_add_f64 %0, %0, %1" : "+v"(a[2 * i + 1]) : "v"(c));
            }
            if constexpr (OP == 5) asm volatile("ds_read_b128 %0, %1" : "=v"(q[i]) : "v"(addr * 4));
            if constexpr (OP == 6) asm volatile("ds_swizzle_b32 %0, %1 offset:swizzle(SWAP,1)" : "=v"(n[i]) : "v"(n[(i + 1) % U]));
        }
        asm volatile("s_waitcnt lgkmcnt(0)");
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    double s = 0;
#pragma unroll
    for (int i = 0; i < U; ++i) s += n[i] + a[2 * i] + a[2 * i + 1] + q[i].x;
    if (s == 12345.678) out[0] = s;
    if ((threadIdx.x & 63) == 0) { atomicAdd(clk + 0, t1 - t0); atomicAdd(clk + 2, 1ull); }
}

template <int OP>
void run(const char* name, double* d, unsigned long long* clk, int n_cu) {
    printf("%-34s", name);
    for (int w : {1, 2, 3}) {
        const int blocks = n_cu * w;
        const size_t lds = (size_t)160 * 1024 / w - 1024;
        hipFuncSetAttribute(reinterpret_cast<const void*>(&k<OP>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), lds, 0, d, clk, 10, 1);
        hipDeviceSynchronize();
        hipMemset(clk, 0, 64);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), lds, 0, d, clk, ITER, 1);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        unsigned long long h[3]; hipMemcpy(h, clk, sizeof(h), hipMemcpyDeviceToHost);
        // wall ns per loop body (U ops of the kind) per CU, i.e. for the 4 * w waves of a CU together; and per single op per CU
        const double ns_body_cu = ms * 1e6 / ITER;
        printf(" | w=%d: %7.1f ns/iter = %6.2f ns per op-slot per CU (all %2d waves), %6.1f cyc/iter/wave", w, ns_body_cu, ns_body_cu / (U * 4 * w), 4 * w,
               (double)h[0] / (double)h[2] / ITER);
    }
    printf("\n");
}

int main() {
    double* d; CHECK(hipMalloc(&d, 64));
    unsigned long long* clk; CHECK(hipMalloc(&clk, 64));
    hipDeviceProp_t p; CHECK(hipGetDeviceProperties(&p, 0));
    printf("device %s, %d CUs; loop body = %d ops of the kind per wave; s_memtime cycles at 100 MHz x ?\n", p.gcnArchName, p.multiProcessorCount, U);
    run<0>("ds_bpermute_b32", d, clk, p.multiProcessorCount);
    run<1>("v_mov_b32_dpp wave_shr:1", d, clk, p.multiProcessorCount);
    run<6>("ds_swizzle_b32", d, clk, p.multiProcessorCount);
    run<5>("ds_read_b128", d, clk, p.multiProcessorCount);
    run<4>("2 v_add_f64 per slot", d, clk, p.multiProcessorCount);
    run<2>("ds_bpermute + 2 v_add_f64", d, clk, p.multiProcessorCount);
    run<3>("ds_bpermute + 4 v_add_f64", d, clk, p.multiProcessorCount);
    return 0;
}
