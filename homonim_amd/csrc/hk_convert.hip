// hk_convert.hip -- data-format edges of the hot path on gfx950: integer / float64 rasters in, typed rasters out.
//
// Reference behaviour:
//   * input : RasterArray.from_rio_dataset reads every band with out_dtype float32 (homonim/raster_array.py:178-188):
//             a plain value conversion (exact for 8/16-bit integers, round-to-nearest for 32-bit integers / float64).
//   * output: RasterArray._convert_array_dtype (homonim/raster_array.py:353-387): promote to a float type that holds
//             the destination range (float32 for <= 16-bit integers, float64 for 32-bit ones), np.round (half to
//             even), np.clip to the destination range, cast; pixels that are nodata in the corrected block (NaN)
//             receive the output nodata value.
// Both are HBM-bound element-wise kernels; they exist so that host<->device traffic is 1-2 B per pixel instead of 4.
#include "hk_kernels.h"

namespace hk {

template <typename T>
__global__ void __launch_bounds__(256) cast_in_kernel(const T* __restrict__ in, long long in_stride, float* __restrict__ out,
                                                      long long out_stride, int height, int width) {
    const int x = (blockIdx.x * blockDim.x + threadIdx.x) * PX;
    if (x >= width) return;
    for (int y = blockIdx.y; y < height; y += gridDim.y) {
        const T* ip = in + (long long)y * in_stride + x;
        float v[PX];
#pragma unroll
        for (int i = 0; i < PX; ++i) v[i] = (float)ip[i];  // rows are padded to a multiple of PX elements
        *reinterpret_cast<float4*>(out + (long long)y * out_stride + x) = make_float4(v[0], v[1], v[2], v[3]);
    }
}

template <typename T>
struct OutTraits;
template <> struct OutTraits<unsigned char>  { static constexpr double lo = 0., hi = 255.; static constexpr bool wide = false, is_float = false; };
template <> struct OutTraits<unsigned short> { static constexpr double lo = 0., hi = 65535.; static constexpr bool wide = false, is_float = false; };
template <> struct OutTraits<short>          { static constexpr double lo = -32768., hi = 32767.; static constexpr bool wide = false, is_float = false; };
template <> struct OutTraits<unsigned int>   { static constexpr double lo = 0., hi = 4294967295.; static constexpr bool wide = true, is_float = false; };
template <> struct OutTraits<int>            { static constexpr double lo = -2147483648., hi = 2147483647.; static constexpr bool wide = true, is_float = false; };
template <> struct OutTraits<float>          { static constexpr double lo = 0., hi = 0.; static constexpr bool wide = false, is_float = true; };
template <> struct OutTraits<double>         { static constexpr double lo = 0., hi = 0.; static constexpr bool wide = true, is_float = true; };

template <typename T>
__global__ void __launch_bounds__(256) cast_out_kernel(const float* __restrict__ in, long long in_stride, T* __restrict__ out,
                                                       long long out_stride, int height, int width, int has_nodata,
                                                       double nodata) {
    using TR = OutTraits<T>;
    const int x = (blockIdx.x * blockDim.x + threadIdx.x) * PX;
    if (x >= width) return;
    const T nd = has_nodata ? (T)nodata : (T)0;
    for (int y = blockIdx.y; y < height; y += gridDim.y) {
        const float4 f4 = *reinterpret_cast<const float4*>(in + (long long)y * in_stride + x);
        const float f[PX] = {f4.x, f4.y, f4.z, f4.w};
        T* op = out + (long long)y * out_stride + x;
#pragma unroll
        for (int i = 0; i < PX; ++i) {
            T o;
            if (f[i] != f[i]) {
                o = has_nodata ? nd : (TR::is_float ? (T)f[i] : (T)0);  // masked pixel -> output nodata
            } else if constexpr (TR::is_float) {
                o = (T)f[i];
            } else if constexpr (TR::wide) {
                double v = rint((double)f[i]);
                v = fmin(fmax(v, TR::lo), TR::hi);
                o = (T)v;
            } else {
                float v = rintf(f[i]);  // np.round: half to even
                v = fminf(fmaxf(v, (float)TR::lo), (float)TR::hi);
                o = (T)v;
            }
            op[i] = o;
        }
    }
}

static dim3 cast_grid(int height, int width) {
    return dim3((width + 256 * PX - 1) / (256 * PX), height < 2048 ? height : 2048);
}

hipError_t launch_cast_in(int dtype, const void* in, long long in_stride, float* out, long long out_stride, int height,
                          int width, hipStream_t stream) {
    const dim3 g = cast_grid(height, width), b(256);
#define HK_CAST_IN(T) HK_LAUNCH(cast_in_kernel<T>, g, b, 0, stream, static_cast<const T*>(in), in_stride, out, out_stride, height, width)
    switch (dtype) {
        case 1: HK_CAST_IN(unsigned char); break;
        case 2: HK_CAST_IN(unsigned short); break;
        case 3: HK_CAST_IN(short); break;
        case 4: HK_CAST_IN(unsigned int); break;
        case 5: HK_CAST_IN(int); break;
        case 6: HK_CAST_IN(double); break;
        default: return hipErrorInvalidValue;
    }
#undef HK_CAST_IN
    return hipGetLastError();
}

hipError_t launch_cast_out(int dtype, const float* in, long long in_stride, void* out, long long out_stride, int height,
                           int width, int has_nodata, double nodata, hipStream_t stream) {
    const dim3 g = cast_grid(height, width), b(256);
#define HK_CAST_OUT(T) HK_LAUNCH(cast_out_kernel<T>, g, b, 0, stream, in, in_stride, static_cast<T*>(out), out_stride, height, width, has_nodata, nodata)
    switch (dtype) {
        case 0: HK_CAST_OUT(float); break;
        case 1: HK_CAST_OUT(unsigned char); break;
        case 2: HK_CAST_OUT(unsigned short); break;
        case 3: HK_CAST_OUT(short); break;
        case 4: HK_CAST_OUT(unsigned int); break;
        case 5: HK_CAST_OUT(int); break;
        case 6: HK_CAST_OUT(double); break;
        default: return hipErrorInvalidValue;
    }
#undef HK_CAST_OUT
    return hipGetLastError();
}

}  // namespace hk
