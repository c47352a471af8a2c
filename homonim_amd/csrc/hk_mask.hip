// hk_mask.hip -- `mask_partial` on a shared grid: KernelModel._full_coverage_mask (homonim/kernel_model.py:375-409).
//
//   mask  = (valid(in) >= 1) & param_ra.mask            param_ra.mask: any of the gain / offset bands is not NaN
//   mask  = cv.erode(mask, ones((kh + 2, kw + 2)), borderType=BORDER_CONSTANT, borderValue=0)
//   RefSpaceModel.apply: parameters outside the mask become NaN before gain * src + offset (:493-503);
//   SrcSpaceModel.fit  : all parameter bands outside the mask become NaN (:526-531).
// Erosion by a full rectangle with a zero border == "the (kh+2) x (kw+2) window count equals its area", evaluated
// separably: row counts (uint16) then column sums.  Optional feature, O(kh + kw) reads per pixel from L1/L2.
#include "hk_kernels.h"

namespace hk {

// pass 1: validity byte of every pixel + its horizontal window count
__global__ void __launch_bounds__(256) mask_rows_kernel(const float* __restrict__ in, int nd_mode, float nodata,
                                                        const float* __restrict__ gain, const float* __restrict__ offset,
                                                        long long stride, int height, int width, int rwe,
                                                        unsigned short* __restrict__ rowcnt) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    if (x >= width) return;
    for (int y = blockIdx.y; y < height; y += gridDim.y) {
        const long long row = (long long)y * stride;
        int c = 0;
        for (int dx = -rwe; dx <= rwe; ++dx) {
            const int xx = x + dx;
            if (xx < 0 || xx >= width) continue;  // zero border
            const float v = in[row + xx], g = gain[row + xx], o = offset[row + xx];
            // mode 3: `in` is the source mask re-projected with `average` -- fully covered pixels only (:399)
            const bool valid = nd_mode == 3 ? (v >= 1.f) : (nd_mode == 0 ? true : (nd_mode == 1 ? !(v != v) : !(v == nodata)));
            c += (valid && (!(g != g) || !(o != o))) ? 1 : 0;
        }
        rowcnt[row + x] = (unsigned short)c;
    }
}

// pass 2: column sums of the row counts == full area -> covered; mask the parameters and/or apply them
__global__ void __launch_bounds__(256) mask_cols_kernel(const unsigned short* __restrict__ rowcnt, long long stride,
                                                        int height, int width, int rhe, int full,
                                                        const float* __restrict__ params, long long band_stride,
                                                        int n_bands, const float* __restrict__ src, float* __restrict__ params_out,
                                                        float* __restrict__ corr_out, unsigned char* __restrict__ mask_out) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    if (x >= width) return;
    const float nan = __int_as_float(0x7fc00000);
    for (int y = blockIdx.y; y < height; y += gridDim.y) {
        int c = 0;
        for (int dy = -rhe; dy <= rhe; ++dy) {
            const int yy = y + dy;
            if (yy >= 0 && yy < height) c += rowcnt[(long long)yy * stride + x];
        }
        const bool covered = c == full;
        const long long off = (long long)y * stride + x;
        if (mask_out) mask_out[off] = covered ? 1 : 0;
        if (params_out)
            for (int b = 0; b < n_bands; ++b) params_out[b * band_stride + off] = covered ? params[b * band_stride + off] : nan;
        if (corr_out) {
            const float g = covered ? params[off] : nan, o = covered ? params[band_stride + off] : nan;
            corr_out[off] = __fadd_rn(__fmul_rn(g, src[off]), o);  // KernelModel.apply (:461)
        }
    }
}

hipError_t launch_partial_mask(const float* in, int nd_mode, float nodata, const float* params, int n_bands,
                               long long band_stride, const float* src, int height, int width, long long stride, int kh,
                               int kw, unsigned short* rowcnt_ws, float* params_out, float* corr_out,
                               unsigned char* mask_out, hipStream_t stream) {
    const dim3 block(256), grid((width + 255) / 256, height < 1024 ? height : 1024);
    const int rhe = kh / 2 + 1, rwe = kw / 2 + 1;  // structuring element (kh + 2) x (kw + 2)
    HK_LAUNCH(mask_rows_kernel, grid, block, 0, stream, in, nd_mode, nodata, params, params + band_stride, stride,
                       height, width, rwe, rowcnt_ws);
    HK_LAUNCH(mask_cols_kernel, grid, block, 0, stream, rowcnt_ws, stride, height, width, rhe,
                       (kh + 2) * (kw + 2), params, band_stride, n_bands, src, params_out, corr_out, mask_out);
    return hipGetLastError();
}

}  // namespace hk
