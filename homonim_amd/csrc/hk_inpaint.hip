// hk_inpaint.hip -- in-painting of the offset band where the kernel models are poor (homonim/kernel_model.py:361-371).
//
// The reference calls rasterio.fill.fillnodata(offset, r2_mask) == GDALFillNodata(max_search_distance = 100,
// smoothing_iterations = 0).  GDAL is not part of /root/reference (rasterio>=1.1, un-pinned) and not installed here,
// so its published algorithm (gdal/alg/rasterfill.cpp) is RESTATED -- parity with GDAL itself is unpinned:
//   * two column scans give, for every pixel, the nearest source pixel (mask != 0) at-or-above it and strictly below
//     it in its own column, carried at most `max_dist` rows (kept here as uint16 row distances);
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
constexpr unsigned short NONE_D = 0xffff;

// source = (r2 > thresh) & (gain > 0) & valid (kernel_model.py:363): one byte per pixel.  NaN parameters (masked pixels,
// degenerate windows) compare false.
__global__ void __launch_bounds__(256) inpaint_flag_kernel(const float* __restrict__ gain, const float* __restrict__ r2,
                                                           float thresh, long long stride, int height, int width,
                                                           unsigned char* __restrict__ flag) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    if (x >= width) return;
    for (int y = blockIdx.y; y < height; y += gridDim.y) {
        const long long i = (long long)y * stride + x;
        flag[i] = ((r2[i] > thresh) && (gain[i] > 0.f)) ? 1 : 0;
    }
}

// Column tables as DISTANCES (uint16): rows up to the nearest source at-or-above (0..max_dist) and down to the nearest
// source strictly below (1..max_dist + 1), NONE_D when there is none in reach; the search reads the source's value from
// the offset plane itself.  One thread per (column, chunk of SCAN_ROWS rows): a source is only visible `max_dist` rows
// away, so a chunk's scan starts `max_dist` rows before (top-down) / after (bottom-up) its first output row with an
// empty state and reproduces the sequential scan exactly.
constexpr int SCAN_ROWS = 256;
__global__ void __launch_bounds__(256) inpaint_scan_kernel(const unsigned char* __restrict__ flag, long long stride, int height,
                                                           int width, int max_dist, unsigned short* __restrict__ top_d,
                                                           unsigned short* __restrict__ bot_d) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    if (x >= width) return;
    const int y0 = blockIdx.y * SCAN_ROWS, y1 = min(y0 + SCAN_ROWS, height);
    // top-down: nearest source at or above each row
    int last_y = NONE_Y;
    for (int y = max(0, y0 - max_dist); y < y1; ++y) {
        const long long i = (long long)y * stride + x;
        if (flag[i]) {
            last_y = y;
        } else if (last_y != NONE_Y && y > max_dist + last_y) {
            last_y = NONE_Y;
        }
        if (y >= y0) top_d[i] = last_y == NONE_Y ? NONE_D : (unsigned short)(y - last_y);
    }
    // bottom-up: nearest source strictly below each row (the state left by the row underneath)
    last_y = NONE_Y;
    for (int y = min(height - 1, y1 + max_dist); y >= y0; --y) {
        const long long i = (long long)y * stride + x;
        if (y < y1) bot_d[i] = last_y == NONE_Y ? NONE_D : (unsigned short)(last_y - y);
        if (flag[i]) {
            last_y = y;
        } else if (last_y != NONE_Y && last_y - y > max_dist) {
            last_y = NONE_Y;
        }
    }
}

// GDAL's QUAD_CHECK compares the squared distance of a candidate with the ROUNDED square of the current distance,
//     if (d2 < qd * qd) { qd = sqrt(d2); ... }          (float64; d2 is an exact integer < 2^15)
// so a candidate at the SAME squared distance n replaces the current source exactly when fl(fl(sqrt(n))^2) > n.  That is a
// property of n alone: tie_kernel tabulates it as a bitmap (TIE_N bits), and the search itself runs on integers -- no
// sqrt and no float64 in the loop, same decisions.
constexpr int TIE_N = 2 * 102 * 102;  // > the largest squared distance the search can meet (100^2 + 101^2)
constexpr int WTAB_N = 100 * 100 + 1;  // weights 1 / qd of the accepted distances (qd <= max_dist = 100)
__global__ void __launch_bounds__(256) tie_kernel(unsigned* __restrict__ tie, double* __restrict__ wtab) {
    const int word = blockIdx.x * blockDim.x + threadIdx.x;
    // the inverse-distance weights GDAL forms as 1.0 / qd with qd = sqrt(n): a table instead of a sqrt and a division
    // per quadrant and target
    for (int n = word; n < WTAB_N; n += gridDim.x * blockDim.x) wtab[n] = n ? 1.0 / sqrt((double)n) : 0.0;
    if (word * 32 >= TIE_N) return;
    unsigned bits = 0;
    for (int b = 0; b < 32; ++b) {
        const double n = (double)(word * 32 + b);
        const double q = sqrt(n);
        if (__dmul_rn(q, q) > n) bits |= 1u << b;
    }
    tie[word] = bits;
}

__device__ __forceinline__ void quad_check(int& qd2, int& qx, int& qy, int tx, int ty, int ox, int oy,
                                           const unsigned* __restrict__ tie) {
    if (ty == NONE_Y) return;
    const int dx = tx - ox, dy = ty - oy;
    const int d2 = dx * dx + dy * dy;
    bool better = d2 < qd2;
    if (d2 == qd2) better = (tie[d2 >> 5] >> (d2 & 31)) & 1u;  // rare
    if (better) qd2 = d2, qx = tx, qy = ty;
}

__global__ void __launch_bounds__(256) inpaint_fill_kernel(const float* __restrict__ offset, const unsigned char* __restrict__ flag,
                                                           long long stride, int height, int width, int max_dist,
                                                           const unsigned short* __restrict__ top_d,
                                                           const unsigned short* __restrict__ bot_d,
                                                           const unsigned* __restrict__ tie,
                                                           const double* __restrict__ wtab, float* __restrict__ filled) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    if (x >= width) return;
    for (int y = blockIdx.y; y < height; y += gridDim.y) {  // grid-stride over rows: blocks taller than 65535 rows are fine
    const long long row = (long long)y * stride;
    const long long i = row + x;
    float out = offset[i];
    if (!flag[i]) {
        const int none2 = (max_dist + 1) * (max_dist + 1);  // qd = max_dist + 1: "nothing found yet" (a perfect square: no tie)
        int qd2[4] = {none2, none2, none2, none2};
        int qx[4] = {0, 0, 0, 0}, qy[4] = {0, 0, 0, 0};
        // Steps are taken in groups that end where GDAL re-derives its search bound (after steps 4, 8, 12, ...): the
        // bound is constant inside a group, so all of the group's table look-ups (4 per step) are issued before the
        // checks, which then run in the original order (ascending step; left quadrants before right ones).
        // The look-ups of the NEXT group are issued before the checks of the current one (their latency hides behind the
        // checks; a group that turns out not to be needed costs four cached loads per step and nothing else).
        constexpr int G = 5;  // longest group (steps 0..4)
        unsigned short lt[G], lb[G], rt[G], rb[G], nlt[G], nlb[G], nrt[G], nrb[G];
        auto fetch = [&](int first_step, unsigned short (&a)[G], unsigned short (&b)[G], unsigned short (&c)[G],
                         unsigned short (&d)[G]) {
#pragma unroll
            for (int k = 0; k < G; ++k) {
                const int step = first_step + k;  // steps beyond the bound are clamped into the row and never checked
                const long long li = row + max(0, x - step), ri = row + min(width - 1, x + step);
                a[k] = top_d[li], b[k] = bot_d[li], c[k] = top_d[ri], d[k] = bot_d[ri];
            }
        };
        int this_max = max_dist;
        int first = 0;
        fetch(0, lt, lb, rt, rb);
        while (first <= this_max) {
            const int last = min(this_max, first == 0 ? 4 : first + 3);
            fetch(last + 1, nlt, nlb, nrt, nrb);
#pragma unroll
            for (int k = 0; k < G; ++k) {
                const int step = first + k;
                if (step <= last) {
                    const int lx = max(0, x - step), rx = min(width - 1, x + step);
                    quad_check(qd2[0], qx[0], qy[0], lx, lt[k] == NONE_D ? NONE_Y : y - (int)lt[k], x, y, tie);  // top left
                    quad_check(qd2[1], qx[1], qy[1], lx, lb[k] == NONE_D ? NONE_Y : y + (int)lb[k], x, y, tie);  // bottom left
                    if (step != 0) {
                        quad_check(qd2[2], qx[2], qy[2], rx, rt[k] == NONE_D ? NONE_Y : y - (int)rt[k], x, y, tie);  // top right
                        quad_check(qd2[3], qx[3], qy[3], rx, rb[k] == NONE_D ? NONE_Y : y + (int)rb[k], x, y, tie);  // bottom right
                    }
                }
            }
            // no farther column can beat every quadrant's current distance: floor(max qd) = floor(sqrt(max qd2))
            if (last >= 4 && (last & 3) == 0)
                this_max = (int)floor(sqrt((double)max(max(qd2[0], qd2[1]), max(qd2[2], qd2[3]))));
            first = last + 1;
#pragma unroll
            for (int k = 0; k < G; ++k) lt[k] = nlt[k], lb[k] = nlb[k], rt[k] = nrt[k], rb[k] = nrb[k];
        }
        double wsum = 0.0, vsum = 0.0;
        bool has = false;
        for (int q = 0; q < 4; ++q) {
            if (qd2[q] <= max_dist * max_dist) {  // qd <= max_dist
                const double w = wtab[qd2[q]];
                has = w != 0.0;
                wsum += w;
                vsum += (double)offset[(long long)qy[q] * stride + qx[q]] * w;
            }
        }
        if (has) out = (float)(vsum / wsum);
    }
    filled[i] = out;
    }
}

// workspace: two uint16 distance tables + source flags (1 byte per pixel) + the tie bitmap + the weight table
size_t inpaint_workspace_bytes(int height, long long stride) { return (size_t)height * stride * 5 + 1024 + TIE_N / 8 + 256 + WTAB_N * 8 + 256; }

// the workspace's source-flag plane: the fit kernel can write it itself (FitArgs::flag), then gain / r2 are not needed here
unsigned char* inpaint_flag_plane(void* workspace, int height, long long stride) {
    return reinterpret_cast<unsigned char*>(static_cast<unsigned short*>(workspace) + 2 * (size_t)height * stride);
}

hipError_t launch_inpaint_offsets(const float* offset, const float* gain, const float* r2, float thresh, long long stride,
                                  int height, int width, void* workspace, float* filled, hipStream_t stream,
                                  const unsigned char* flag_ready) {
    const size_t plane = (size_t)height * stride;
    unsigned short* top_d = static_cast<unsigned short*>(workspace);
    unsigned short* bot_d = top_d + plane;
    unsigned char* ws_flag = inpaint_flag_plane(workspace, height, stride);
    const unsigned char* flag = flag_ready ? flag_ready : ws_flag;
    unsigned* tie = reinterpret_cast<unsigned*>(ws_flag + (plane + 255) / 256 * 256);
    double* wtab = reinterpret_cast<double*>(tie + (TIE_N / 32 + 64) / 64 * 64);
    hipLaunchKernelGGL(tie_kernel, dim3((TIE_N / 32 + 255) / 256), dim3(256), 0, stream, tie, wtab);
    const int max_dist = 100;  // rasterio.fill.fillnodata default max_search_distance (kernel_model.py:366)
    if (!flag_ready)  // else: the flag plane was written by the fit kernel (FitArgs::flag)
        hipLaunchKernelGGL(inpaint_flag_kernel, dim3((width + 255) / 256, height < 1024 ? height : 1024), dim3(256), 0, stream,
                           gain, r2, thresh, stride, height, width, ws_flag);
    hipLaunchKernelGGL(inpaint_scan_kernel, dim3((width + 255) / 256, (height + SCAN_ROWS - 1) / SCAN_ROWS), dim3(256), 0,
                       stream, flag, stride, height, width, max_dist, top_d, bot_d);
    hipLaunchKernelGGL(inpaint_fill_kernel, dim3((width + 255) / 256, height < 65535 ? height : 65535), dim3(256), 0, stream, offset, flag, stride,
                       height, width, max_dist, top_d, bot_d, tie, wtab, filled);
    return hipGetLastError();
}

}  // namespace hk
