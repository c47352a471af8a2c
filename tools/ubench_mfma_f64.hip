// ubench_mfma_f64.hip -- what does v_mfma_f64_16x16x4_f64 cost on MI355X?  (could a banded-ones MFMA replace
// the float64 horizontal stage of the fused kernel -- per wave-row of 256 px and per window sum 9 float64 adds + 8 DPP moves?)
// Times a loop of independent MFMAs (4 accumulator sets per wave) and, beside it, a loop of float64 adds, at 1 / 2 / 4 waves per
// SIMD; prints wave-instructions per second per SIMD and cycles per instruction at the shader clock given (MHz).
//   hipcc --offload-arch=gfx950 -O3 -o tools/ubench_mfma_f64 tools/ubench_mfma_f64.hip ; ./tools/ubench_mfma_f64 [sclk_mhz]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

typedef double d4 __attribute__((ext_vector_type(4)));

__global__ void __launch_bounds__(256) mfma_loop(double* out, int iters, double seed) {
    d4 acc[4];
    for (int k = 0; k < 4; ++k) acc[k] = d4{seed, seed, seed, seed};
    double a = seed + threadIdx.x, b = 1.0;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < 4; ++k) acc[k] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[k], 0, 0, 0);
    }
    double s = 0;
    for (int k = 0; k < 4; ++k) s += acc[k].x + acc[k].y + acc[k].z + acc[k].w;
    if (s == 123.456) out[threadIdx.x] = s;
}

__global__ void __launch_bounds__(256) dadd_loop(double* out, int iters, double seed) {
    double acc[16];
    for (int k = 0; k < 16; ++k) acc[k] = seed + k;
    const double b = seed * 0.5;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < 16; ++k) acc[k] = __dadd_rn(acc[k], b);
    }
    double s = 0;
    for (int k = 0; k < 16; ++k) s += acc[k];
    if (s == 123.456) out[threadIdx.x] = s;
}

int main(int argc, char** argv) {
    const double mhz = argc > 1 ? atof(argv[1]) : 2400.0;
    double* out;
    hipMalloc(&out, 4096);
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    const int cus = 256, simds = cus * 4;
    for (int waves_per_simd : {1, 2, 4}) {
        const int blocks = cus * waves_per_simd;  // 256 threads = 4 waves = one per SIMD of a CU
        for (int which = 0; which < 2; ++which) {
            const int iters = which == 0 ? 20000 : 100000, per_iter = which == 0 ? 4 : 16;
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0);
                if (which == 0) hipLaunchKernelGGL(mfma_loop, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0);
                else hipLaunchKernelGGL(dadd_loop, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
            }
            float ms = 0;
            hipEventElapsedTime(&ms, e0, e1);
            const double instr_per_simd = (double)iters * per_iter * waves_per_simd;
            const double ns = ms * 1e6 / instr_per_simd;
            printf("%-26s %d wave(s) per SIMD: %7.2f ns per wave-instruction per SIMD = %6.1f cycles at %.0f MHz%s\n",
                   which == 0 ? "v_mfma_f64_16x16x4_f64" : "v_add_f64", waves_per_simd, ns, ns * mhz * 1e-3, mhz,
                   which == 0 ? "  (2048 flop each)" : "");
        }
    }
    (void)simds;
    return 0;
}
