// hk_inpaint.hip -- in-painting of the offset band where the kernel models are poor (homonim/kernel_model.py:361-371).
//
// The reference calls rasterio.fill.fillnodata(offset, r2_mask) == GDALFillNodata(max_search_distance = 100,
// smoothing_iterations = 0).  GDAL is not part of /root/reference (rasterio>=1.1, un-pinned) and not installed here,
// so its published algorithm (gdal/alg/rasterfill.cpp) is RESTATED -- parity with GDAL itself is unpinned:
//   * two column scans give, for every pixel, the nearest source pixel (mask != 0) at-or-above it and strictly below
//     it in its own column, carried at most `max_dist` rows;
//   * a target pixel (mask == 0) steps left and right one column at a time (0..max_dist) and keeps, per quadrant
//     (top-left and bottom-left include the pixel's own column, the right quadrants do not), the closest source found
//     through those column tables -- strict `<` on the squared distance, so the first one met wins ties;
//   * value = sum(v_q / d_q) / sum(1 / d_q) over the quadrants with d_q <= max_dist, in float64, cast to float32;
//     targets without any source keep their value.  Filled pixels never act as sources.
// GDAL runs this sequentially line by line; every target is independent given the column tables, so here it is one
// thread per (column, row chunk) for the scans and one thread per pixel for the search.
#include "hk_kernels.h"

namespace hk {

constexpr int NONE_Y = 0x7fffffff;

// source = (r2 > thresh) & (gain > 0) & valid (kernel_model.py:363); valid <=> gain plane holds a parameter or NaN from
// a degenerate window -- the validity itself travels as the non-NaN-ness of `valid_ref` (the corrected/gain planes are
// NaN outside the mask by construction, but degenerate windows can be NaN inside it, so the mask is passed explicitly).
// One thread per (column, chunk of SCAN_ROWS rows).  A source is only visible `max_dist` rows away, so a chunk's scan
// starts `max_dist` rows before (top-down) / after (bottom-up) its first output row with an empty state and reproduces
// the sequential scan exactly, while the grid has (height / SCAN_ROWS) times more threads than columns.
constexpr int SCAN_ROWS = 128;
__global__ void __launch_bounds__(256) inpaint_scan_kernel(const float* __restrict__ offset, const float* __restrict__ gain,
                                                           const float* __restrict__ r2, float thresh, long long stride,
                                                           int height, int width, int max_dist, int* __restrict__ top_y,
                                                           float* __restrict__ top_v, int* __restrict__ bot_y,
                                                           float* __restrict__ bot_v) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    if (x >= width) return;
    const int y0 = blockIdx.y * SCAN_ROWS, y1 = min(y0 + SCAN_ROWS, height);
    // top-down: nearest source at or above each row
    int last_y = NONE_Y;
    float last_v = 0.f;
    for (int y = max(0, y0 - max_dist); y < y1; ++y) {
        const long long i = (long long)y * stride + x;
        const bool srcpx = (r2[i] > thresh) && (gain[i] > 0.f);  // NaN parameters (masked pixels) compare false
        if (srcpx) {
            last_y = y;
            last_v = offset[i];
        } else if (last_y != NONE_Y && y > max_dist + last_y) {
            last_y = NONE_Y;
        }
        if (y >= y0) {
            top_y[i] = last_y;
            top_v[i] = last_v;
        }
    }
    // bottom-up: nearest source strictly below each row (the state left by the row underneath)
    last_y = NONE_Y;
    last_v = 0.f;
    for (int y = min(height - 1, y1 - 1 + max_dist + 1); y >= y0; --y) {
        const long long i = (long long)y * stride + x;
        if (y < y1) {
            bot_y[i] = last_y;
            bot_v[i] = last_v;
        }
        const bool srcpx = (r2[i] > thresh) && (gain[i] > 0.f);
        if (srcpx) {
            last_y = y;
            last_v = offset[i];
        } else if (last_y != NONE_Y && last_y - y > max_dist) {
            last_y = NONE_Y;
        }
    }
}

__device__ __forceinline__ void quad_check(double& qd, double& qv, int tx, int ty, int ox, int oy, float tv) {
    if (ty == NONE_Y) return;
    const double dx = (double)tx - (double)ox, dy = (double)ty - (double)oy;
    const double d2 = dx * dx + dy * dy;
    if (d2 < qd * qd) {
        qd = sqrt(d2);
        qv = (double)tv;
    }
}

__global__ void __launch_bounds__(256) inpaint_fill_kernel(const float* __restrict__ offset, const float* __restrict__ gain,
                                                           const float* __restrict__ r2, float thresh, long long stride,
                                                           int height, int width, int max_dist, const int* __restrict__ top_y,
                                                           const float* __restrict__ top_v, const int* __restrict__ bot_y,
                                                           const float* __restrict__ bot_v, float* __restrict__ filled) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y;
    if (x >= width) return;
    const long long row = (long long)y * stride;
    const long long i = row + x;
    float out = offset[i];
    const bool srcpx = (r2[i] > thresh) && (gain[i] > 0.f);
    if (!srcpx) {
        double qd[4], qv[4] = {0.0, 0.0, 0.0, 0.0};
        for (int q = 0; q < 4; ++q) qd[q] = (double)max_dist + 1.0;
        // Steps are taken in groups that end where GDAL re-derives its search bound (after steps 4, 8, 12, ...): the
        // bound is constant inside a group, so all of the group's table look-ups (8 per step) are issued before the
        // checks, which then run in the original order (ascending step; left quadrants before right ones).
        int this_max = max_dist;
        int first = 0;
        while (first <= this_max) {
            const int last = min(this_max, first == 0 ? 4 : first + 3);
            constexpr int G = 5;  // longest group (steps 0..4)
            int ly[G][2], ry[G][2];
            float lv[G][2], rv[G][2];
#pragma unroll
            for (int k = 0; k < G; ++k) {
                const int step = first + k;
                if (step <= last) {
                    const long long li = row + max(0, x - step), ri = row + min(width - 1, x + step);
                    ly[k][0] = top_y[li], lv[k][0] = top_v[li], ly[k][1] = bot_y[li], lv[k][1] = bot_v[li];
                    ry[k][0] = top_y[ri], rv[k][0] = top_v[ri], ry[k][1] = bot_y[ri], rv[k][1] = bot_v[ri];
                }
            }
#pragma unroll
            for (int k = 0; k < G; ++k) {
                const int step = first + k;
                if (step <= last) {
                    const int lx = max(0, x - step), rx = min(width - 1, x + step);
                    quad_check(qd[0], qv[0], lx, ly[k][0], x, y, lv[k][0]);  // top left (own column, own row incl.)
                    quad_check(qd[1], qv[1], lx, ly[k][1], x, y, lv[k][1]);  // bottom left
                    if (step != 0) {
                        quad_check(qd[2], qv[2], rx, ry[k][0], x, y, rv[k][0]);  // top right
                        quad_check(qd[3], qv[3], rx, ry[k][1], x, y, rv[k][1]);  // bottom right
                    }
                }
            }
            // no farther column can beat every quadrant's current distance
            if (last >= 4 && (last & 3) == 0) this_max = (int)floor(fmax(fmax(qd[0], qd[1]), fmax(qd[2], qd[3])));
            first = last + 1;
        }
        double wsum = 0.0, vsum = 0.0;
        bool has = false;
        for (int q = 0; q < 4; ++q) {
            if (qd[q] <= (double)max_dist) {
                const double w = 1.0 / qd[q];
                has = w != 0.0;
                wsum += w;
                vsum += qv[q] * w;
            }
        }
        if (has) out = (float)(vsum / wsum);
    }
    filled[i] = out;
}

size_t inpaint_workspace_bytes(int height, long long stride) { return (size_t)height * stride * 16; }

hipError_t launch_inpaint_offsets(const float* offset, const float* gain, const float* r2, float thresh, long long stride,
                                  int height, int width, void* workspace, float* filled, hipStream_t stream) {
    const size_t plane = (size_t)height * stride;
    int* top_y = static_cast<int*>(workspace);
    float* top_v = reinterpret_cast<float*>(top_y + plane);
    int* bot_y = reinterpret_cast<int*>(top_v + plane);
    float* bot_v = reinterpret_cast<float*>(bot_y + plane);
    const int max_dist = 100;  // rasterio.fill.fillnodata default max_search_distance (kernel_model.py:366)
    hipLaunchKernelGGL(inpaint_scan_kernel, dim3((width + 255) / 256, (height + SCAN_ROWS - 1) / SCAN_ROWS), dim3(256), 0,
                       stream, offset, gain, r2, thresh,
                       stride, height, width, max_dist, top_y, top_v, bot_y, bot_v);
    hipLaunchKernelGGL(inpaint_fill_kernel, dim3((width + 255) / 256, height), dim3(256), 0, stream, offset, gain, r2,
                       thresh, stride, height, width, max_dist, top_y, top_v, bot_y, bot_v, filled);
    return hipGetLastError();
}

}  // namespace hk
