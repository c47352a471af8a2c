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

#include <stdlib.h>

#define HK_SQ(v) __mul24((v), (v))  // squares of column distances <= 100: the full-rate 24-bit multiply (v_mul_lo_u32 runs at a quarter)

#include <type_traits>

namespace hk {


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

// Column table: per pixel two SQUARED row distances in one 32-bit word (round 5; rounds 1-4: the distances as two bytes) -- low
// half: rows up to the nearest source at-or-above (0..max_dist), squared; high half: rows down to the nearest source strictly
// below (1..max_dist + 1), squared; NONE_SQ when there is none in reach (max_dist = 100, rasterio's default).  A candidate test
// of the search then needs no multiplication and no separate row-distance field: its key is (entry half << 15) + a per-step
// constant (fill_one), five instructions instead of seven; the row distance comes back as an exact square root at the finish.
// The search reads the source's value from the offset plane itself.  GDAL carries both as the state of two sequential column scans; "the nearest source within reach" is
// the same thing without the sequence: the flags of a column are packed into 64-row bit words (inpaint_bits_kernel: 64
// independent byte loads per thread), and every pixel finds its two distances with clz / ctz on at most three words per
// direction (inpaint_table_kernel: one thread per column and word, no memory access inside its 64-row loop).  The
// sequential form (one thread per column and 64-row chunk, 100 rows of run-in per direction) took 1.15 ms per 16384^2
// band whatever its chunk height, load width or unrolling: its loads sit behind data-dependent state updates.
constexpr unsigned NONE_B = 0xffu;      // (row-distance byte of the table kernel's sweeps: none in reach)
constexpr unsigned NONE_SQ = 0x7fffu;  // squared-distance half of a table entry: none in reach (beats nothing: > 2 * 101^2)
constexpr int WORD_ROWS = 64;

__global__ void __launch_bounds__(256) inpaint_bits_kernel(const unsigned char* __restrict__ flag, long long stride, int height,
                                                           int width, unsigned long long* __restrict__ bits) {
    // a thread packs FOUR adjacent columns: 64 independent 4-byte loads (the flag plane's rows are 4-byte aligned: stride % 4 == 0)
    // instead of 64 single bytes per column -- a quarter of the load instructions for the same bytes (0.158 -> 0.064 ms per 16384^2 band)
    const int x = (blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (x >= width) return;
    const int y0 = blockIdx.y * WORD_ROWS;
    unsigned lo[4] = {0, 0, 0, 0}, hi[4] = {0, 0, 0, 0};
#pragma unroll
    for (int r = 0; r < 32; ++r) {
        const int ya = y0 + r, yb = y0 + 32 + r;
        // rows past the raster are clamped into it and masked out (the loads stay unconditional and independent)
        const unsigned fa = *reinterpret_cast<const unsigned*>(flag + (long long)min(ya, height - 1) * stride + x);
        const unsigned fb = *reinterpret_cast<const unsigned*>(flag + (long long)min(yb, height - 1) * stride + x);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            lo[c] |= (unsigned)(((fa >> (8 * c)) & 0xffu) != 0 && ya < height) << r;
            hi[c] |= (unsigned)(((fb >> (8 * c)) & 0xffu) != 0 && yb < height) << r;
        }
    }
#pragma unroll
    for (int c = 0; c < 4; ++c)
        if (x + c < width) bits[(long long)blockIdx.y * stride + x + c] = ((unsigned long long)hi[c] << 32) | lo[c];
}

__global__ void __launch_bounds__(256) inpaint_table_kernel(const unsigned long long* __restrict__ bits, long long stride,
                                                            int height, int width, int max_dist,
                                                            unsigned* __restrict__ tb) {
    // a thread makes the entries of TWO adjacent columns and stores them as one 8-byte pair per row (rows are padded: stride % 4 == 0;
    // the table starts 256-byte aligned)
    const int x = (blockIdx.x * blockDim.x + threadIdx.x) * 2;
    if (x >= width) return;
    const int wb = blockIdx.y, n_words = (height + WORD_ROWS - 1) / WORD_ROWS;
    const int y0 = wb * WORD_ROWS, rows = min(WORD_ROWS, height - y0);
    unsigned long long wm2[2], wm1[2], w0[2], wp1[2], wp2[2];
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        auto word = [&](int w) { return (w >= 0 && w < n_words) ? bits[(long long)w * stride + x + c] : 0ull; };  // x + 1 < stride
        wm2[c] = word(wb - 2), wm1[c] = word(wb - 1), w0[c] = word(wb), wp1[c] = word(wb + 1), wp2[c] = word(wb + 2);
    }
    // Both distances are running counters along the column: U(b) = 0 where bit b is set, else U(b - 1) + 1 (rows up to the nearest
    // source at or above), D(b) = 1 where bit b + 1 is set, else D(b + 1) + 1 (rows down to the nearest source strictly below); the
    // words above / below only seed them.  An ascending sweep leaves the 64 U bytes packed in registers, a descending sweep adds D
    // and stores -- two or three integer operations per row and column instead of 64-bit clz / ffs with three-way selects
    // (0.247 -> 0.157 ms per 16384^2 band).
    constexpr int BIG = 1 << 20;
    unsigned upk[2][WORD_ROWS / 4];
    int dseed[2];
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        int u = wm1[c] ? __clzll((long long)wm1[c]) : (wm2[c] ? 64 + __clzll((long long)wm2[c]) : BIG);  // U(-1)
#pragma unroll
        for (int q = 0; q < WORD_ROWS / 4; ++q) {
            unsigned pk = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int bb = 4 * q + k;
                u = ((w0[c] >> bb) & 1ull) ? 0 : min(u + 1, BIG);
                pk |= (unsigned)(u <= max_dist ? u : (int)NONE_B) << (8 * k);
            }
            upk[c][q] = pk;
        }
        dseed[c] = wp1[c] ? 1 + (__ffsll((long long)wp1[c]) - 1) : (wp2[c] ? 65 + (__ffsll((long long)wp2[c]) - 1) : BIG);  // D(63)
    }
    int d[2] = {dseed[0], dseed[1]};
#pragma unroll
    for (int bb = WORD_ROWS - 1; bb >= 0; --bb) {
        unsigned e[2];
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            if (bb < WORD_ROWS - 1) d[c] = ((w0[c] >> (bb + 1)) & 1ull) ? 1 : min(d[c] + 1, BIG);
            const unsigned ub = (upk[c][bb >> 2] >> (8 * (bb & 3))) & 0xffu;
            const unsigned usq = ub == NONE_B ? NONE_SQ : (unsigned)HK_SQ((int)ub);
            const unsigned dsq = d[c] <= max_dist + 1 ? (unsigned)HK_SQ(d[c]) : NONE_SQ;
            e[c] = (dsq << 16) | usq;
        }
        // the second column may lie in the row padding (odd width): its word of `bits` was never written, its entry is never read
        if (bb < rows) *reinterpret_cast<uint2*>(tb + (long long)(y0 + bb) * stride + x) = make_uint2(e[0], e[1]);
    }
}

// GDAL's QUAD_CHECK compares the squared distance of a candidate with the ROUNDED square of the current distance,
//     if (d2 < qd * qd) { qd = sqrt(d2); ... }          (float64; d2 is an exact integer < 2^15)
// so a candidate at the SAME squared distance n replaces the current source exactly when fl(fl(sqrt(n))^2) > n.  That is a
// property of n alone: tie_kernel tabulates it as a bitmap (TIE_N bits), and the search itself runs on integers -- no
// sqrt and no float64 in the loop, same decisions.
constexpr int TIE_N = 2 * 102 * 102;  // > the largest squared distance the search can meet (100^2 + 101^2)
constexpr int FILL_MAX_DIST = 100;  // rasterio.fill.fillnodata default max_search_distance (kernel_model.py:366)
constexpr int WTAB_N = FILL_MAX_DIST * FILL_MAX_DIST + 1;  // weights 1 / qd of the accepted distances (qd <= max_dist = 100)
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

// floor(sqrt(n)) for 0 <= n < 2^23 (float32 sqrt is correctly rounded and n is exact in float32; the two corrections cost
// nothing and make the result independent of that argument)
__device__ __forceinline__ int isqrt_floor(int n) {
    int r = (int)__fsqrt_rn((float)n);
    if (r * r > n) --r;
    if ((r + 1) * (r + 1) <= n) ++r;
    return r;
}

template <bool INTERIOR = false>
__device__ __forceinline__ void fill_load_d0(const unsigned* __restrict__ trow, int x, int width, unsigned (&d0)[10]) {
    struct __attribute__((packed, aligned(4))) W10 { unsigned w[10]; };
    if (INTERIOR || (x - 4 >= 0 && x + 5 < width)) {
        const W10 v = *reinterpret_cast<const W10*>(trow + x - 4);
#pragma unroll
        for (int j = 0; j < 10; ++j) d0[j] = v.w[j];
    } else {  // GDAL's clamp: it re-checks the edge column
#pragma unroll
        for (int j = 0; j < 10; ++j) d0[j] = trow[j < 4 ? max(0, x - 4 + j) : min(width - 1, x - 4 + j)];
    }
}

// The search of ONE target pixel (x, y): GDAL's quadrant search through the column tables, then the inverse-distance mean of the
// quadrants' sources.  Returns the filled value (the pixel's own value if no source is in reach).
// INTERIOR (wave-uniform, tiled kernel): no step of any search of the wave can reach the raster's edge columns -- the clamps,
// their per-lane squares and the entry-by-entry table reads fall away.
template <bool INTERIOR = false>
__device__ __forceinline__ float fill_one(int x, int y, long long row, const float* __restrict__ offset, long long stride, int width,
                                          int max_dist, const unsigned* __restrict__ tb, const unsigned* __restrict__ tie,
                                          const double* __restrict__ wtab) {
        const long long i = row + x;
        float out = offset[i];  // a target without any source in reach keeps its value
        const int none2 = (max_dist + 1) * (max_dist + 1);  // qd = max_dist + 1: "nothing found yet" (a perfect square: no tie)
        // Per quadrant the search state is two KEYS (round 4): (squared distance << 15) | (column distance << 8).  The smallest key
        // is the FIRST candidate met at the best distance (steps ascend, so among equal distances the smallest column distance came
        // first), the smallest key with the low 15 bits inverted is the LAST one: two unsigned minima per candidate instead of two
        // compares and three selects.  A candidate that does not beat "nothing found yet" leaves the initial keys in place; quadrants
        // whose best distance exceeds max_dist are dropped at the end as before.  (Round 5: the table holds SQUARED row distances, so a
        // candidate's key is one shift-add of its table half and the step's constant; the row distance of the winner is the exact
        // square root of what is left of its squared distance at the finish.)
        constexpr unsigned SRC_BITS = 15u, SRC_MASK = (1u << SRC_BITS) - 1u;
        unsigned kf[4], kl[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) kf[q] = kl[q] = ((unsigned)none2 << SRC_BITS) | SRC_MASK;
        const unsigned* __restrict__ trow = tb + row;
        // GDAL's QUAD_CHECK on squared integer distances.  `sq` is the column table's half word: NONE_SQ (no source in reach) is more
        // than any squared distance the search accepts and more than the initial (max_dist + 1)^2, so it never wins.
        // A candidate at the SAME squared distance replaces the holder iff GDAL's float comparison says so for that distance (the
        // `tie` bit of c) -- a property of c alone: with the bit set the last candidate met at the best distance wins, without it
        // the first.  So the search keeps both, branch-free, and the bit is looked up once per quadrant at the end instead of behind
        // a divergent branch in every one of the 34 candidate tests.
        // dxk = (dx^2 << 15) | (dx << 8): the step's share of the key (wave-uniform away from the raster's edge columns)
        auto consider = [&](int q, unsigned sq, unsigned dxk) {
            const unsigned key = (sq << SRC_BITS) + dxk;
            kf[q] = min(kf[q], key);
            kl[q] = min(kl[q], key ^ SRC_MASK);
        };
        auto worst_qd2 = [&]() { return (int)(max(max(kf[0], kf[1]), max(kf[2], kf[3])) >> SRC_BITS); };
        // Steps are taken in groups that end where GDAL re-derives its search bound (after steps 4, 8, 12, ...): the
        // bound is constant inside a group, so all of the group's table look-ups are issued before the checks, which then
        // run in the original order (ascending step; left quadrants before right ones).  The look-ups of the NEXT group are
        // issued before the checks of the current one (their latency hides behind the checks; a group that turns out not to be
        // needed costs two cached loads and nothing else).
        // Round 3: a group's table words are consecutive 16-bit entries of one row, so they are fetched by two wide loads
        // (2-byte aligned; steps 0..4: the ten entries around x as 16 + 4 bytes; later groups: 8 bytes per side) instead of
        // ten / eight 2-byte gathers -- the search was bound by the look-ups' issue and latency, not by its arithmetic.
        // Lanes whose group reaches past the raster's edge columns assemble the same words entry by entry with GDAL's clamp
        // (it re-checks the edge column).
        {
            struct __attribute__((packed, aligned(4))) W4 { unsigned w[4]; };
            // entry j of a run of table words: low half = squared distance up, high half = squared distance down
            auto up = [](const unsigned* d, int j) { return d[j] & 0xffffu; };
            auto dn = [](const unsigned* d, int j) { return d[j] >> 16; };
            unsigned d0[10];                 // entries 0..9 <-> columns x - 4 .. x + 5 (steps 0..4: left step k = entry 4 - k, right = 4 + k)
            fill_load_d0<INTERIOR>(trow, x, width, d0);
            // later groups (steps first .. first + 3): left entries 0..3 <-> columns x - first - 3 .. x - first (step first + k = entry
            // 3 - k), right entries 0..3 <-> columns x + first .. x + first + 3 (step first + k = entry k)
            auto fetch4 = [&](int first, unsigned (&l)[4], unsigned (&r)[4]) {
                if (INTERIOR || (x - first - 3 >= 0 && x + first + 3 < width)) {
                    const W4 lv = *reinterpret_cast<const W4*>(trow + x - first - 3);
                    const W4 rv = *reinterpret_cast<const W4*>(trow + x + first);
#pragma unroll
                    for (int j = 0; j < 4; ++j) l[j] = lv.w[j], r[j] = rv.w[j];
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        l[j] = trow[max(0, x - first - 3 + j)];
                        r[j] = trow[min(width - 1, x + first + j)];
                    }
                }
            };
            unsigned nl[4], nr[4];
            fetch4(5, nl, nr);
            int this_max = max_dist;
            {   // steps 0 .. 4
                const int last = min(this_max, 4);
#pragma unroll
                for (int k = 0; k < 5; ++k) {
                    if (k <= last) {
                        const int dl = INTERIOR ? k : min(k, x), dr = INTERIOR ? k : min(k, width - 1 - x);  // clamped columns: the distance to the edge column
                        const unsigned dlk = ((unsigned)HK_SQ(dl) << SRC_BITS) | ((unsigned)dl << 8);
                        const unsigned drk = ((unsigned)HK_SQ(dr) << SRC_BITS) | ((unsigned)dr << 8);
                        consider(0, up(d0, 4 - k), dlk);  // top left
                        consider(1, dn(d0, 4 - k), dlk);  // bottom left
                        if (k != 0) {
                            consider(2, up(d0, 4 + k), drk);  // top right
                            consider(3, dn(d0, 4 + k), drk);  // bottom right
                        }
                    }
                }
                // no farther column can beat every quadrant's current distance: floor(max qd) = floor(sqrt(max qd2))
                if (last == 4) this_max = isqrt_floor(worst_qd2());
            }
            int first = 5;
            while (first <= this_max) {
                // The whole group is tested even when the bound falls inside it: a column farther than floor(sqrt(worst
                // quadrant distance^2)) cannot beat or tie any quadrant (its squared column distance alone exceeds every best),
                // so GDAL's bound only ever decides when to STOP, and a per-step predicate (an exec-mask round trip per step) buys
                // nothing.  Groups start at 5, 9, ..., 97: no step beyond max_dist = 100.
                const int last = first + 3;
                const unsigned cl[4] = {nl[0], nl[1], nl[2], nl[3]}, cr[4] = {nr[0], nr[1], nr[2], nr[3]};
                fetch4(first + 4, nl, nr);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int step = first + k;
                    {
                        const int dl = INTERIOR ? step : min(step, x), dr = INTERIOR ? step : min(step, width - 1 - x);
                        const unsigned dlk = ((unsigned)HK_SQ(dl) << SRC_BITS) | ((unsigned)dl << 8);
                        const unsigned drk = ((unsigned)HK_SQ(dr) << SRC_BITS) | ((unsigned)dr << 8);
                        consider(0, up(cl, 3 - k), dlk);
                        consider(1, dn(cl, 3 - k), dlk);
                        consider(2, up(cr, k), drk);
                        consider(3, dn(cr, k), drk);
                    }
                }
                this_max = min(this_max, isqrt_floor(worst_qd2()));
                first = last + 1;
            }
        }
        // The four quadrants' weights and source values are fetched WITHOUT branches -- a quadrant without a source in reach reads
        // entry 0 of the weight table (0.0) and the pixel's own value, and its terms are selected away -- so that the eight look-ups
        // go out together: as `if (found) { load; load; accumulate }` per quadrant they made four dependent round trips at the end of
        // every search (profiles/r03_fill_tile.txt).  The sums run in the same order over the same terms.
        double w4[4];
        float v4[4];
        bool ok4[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int qd2 = (int)(kf[q] >> SRC_BITS);
            ok4[q] = qd2 <= max_dist * max_dist;  // qd <= max_dist
            const int cq = ok4[q] ? qd2 : 0;
            const unsigned src = ((tie[cq >> 5] >> (cq & 31)) & 1u) ? ((kl[q] & SRC_MASK) ^ SRC_MASK) : (kf[q] & SRC_MASK);  // (the 1.3 KB bitmap stays in cache)
            const int dx = ok4[q] ? (int)(src >> 8) : 0;
            const int dy = (int)__fsqrt_rn((float)(cq - dx * dx));  // exact: a perfect square <= 101^2 (0 for a quadrant that is dropped)
            const int sx = q < 2 ? x - dx : x + dx, sy = (q & 1) ? y + dy : y - dy;
            w4[q] = wtab[cq];
            v4[q] = offset[(long long)sy * stride + sx];
        }
        double wsum = 0.0, vsum = 0.0;
        bool has = false;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            has = ok4[q] ? (w4[q] != 0.0) : has;
            wsum = ok4[q] ? wsum + w4[q] : wsum;
            vsum = ok4[q] ? vsum + (double)v4[q] * w4[q] : vsum;
        }
        if (has) out = (float)(vsum / wsum);
        return out;
}

// 8 waves per SIMD: the step from 6 to 8 resident waves was worth 8 % of the whole in-painting branch while the search was
// bound by dependent look-ups.  With the column table as one 16-bit word per pixel it is bound by its VALU instructions
// (PMC: busy 90-100 %); hence the lean candidate test below: per quadrant the state is the best squared distance and one packed
// word (column distance << 8 | row distance), a candidate costs ~8 instructions.  (A second copy of the loop for wave-rows
// away from the raster's edges, with scalar column distances and unclamped look-ups, cost 18 more VGPRs than it saved
// instructions.)
#ifndef HK_FILL_WAVES_FULL
#define HK_FILL_WAVES_FULL 8
#endif
template <bool PACK>
__global__ void __launch_bounds__(256, PACK ? 8 : HK_FILL_WAVES_FULL) inpaint_fill_kernel(const float* __restrict__ offset, const unsigned char* __restrict__ flag,
                                                           long long stride, int height, int width, int max_dist,
                                                           const unsigned* __restrict__ tb,
                                                           const unsigned* __restrict__ tie,
                                                           const double* __restrict__ wtab, float* __restrict__ filled) {
    // Only target pixels search.  PACK (moderate failure rates: the targets are a minority scattered over the lanes): the
    // workgroup first passes its sources through and COMPACTS its targets (ballot + a 4-entry prefix in LDS), then thread t
    // searches for target t -- full waves instead of a third of the lanes in every wave (-12 % of the branch at 35 %
    // failures).  When nearly every pixel is a target the packing only costs its barriers (+3 % at 94 %): the host picks.
    __shared__ unsigned short lst[PACK ? 256 : 1];
    __shared__ unsigned wcnt[256 / WAVE];
    const int tid = threadIdx.x, lane = tid & (WAVE - 1), wv = tid / WAVE;
    const int x_own = blockIdx.x * blockDim.x + tid;
    if constexpr (!PACK) {
        if (x_own >= width) return;
    }
    for (int y = blockIdx.y; y < height; y += gridDim.y) {  // grid-stride over rows: blocks taller than 65535 rows are fine
    const long long row = (long long)y * stride;
    bool target = false;
    if (x_own < width) {
        target = !flag[row + x_own];
        if (!target) filled[row + x_own] = offset[row + x_own];  // filled pixels never act as sources: sources pass through
    }
    bool active = target;
    int x = x_own;
    // (without the compaction x is the same in every row of the loop: keep the compiler from hoisting the per-step edge
    // distances and conditions out of it -- 89 VGPRs instead of 52, i.e. spills at 8 waves per SIMD)
    asm volatile("" : "+v"(x));
    if constexpr (PACK) {
        const unsigned long long bal = __ballot(target);
        if (lane == 0) wcnt[wv] = (unsigned)__popcll(bal);
        __syncthreads();
        unsigned base = 0, n_targets = 0;
#pragma unroll
        for (int w = 0; w < 256 / WAVE; ++w) {
            base += w < wv ? wcnt[w] : 0u;
            n_targets += wcnt[w];
        }
        if (target) lst[base + (unsigned)__popcll(bal & ((1ull << lane) - 1ull))] = (unsigned short)tid;
        __syncthreads();
        active = tid < (int)n_targets;
        if (active) x = blockIdx.x * blockDim.x + lst[tid];
    }
    if (active) filled[row + x] = fill_one(x, y, row, offset, stride, width, max_dist, tb, tie, wtab);
    }
}

// TILED form of the search (round 3): a wave owns a tile of 64 columns x ROWS rows.  The per-row form above starts a workgroup per
// 256-pixel row piece, and each lives for three dependent memory round trips (flag -> compaction behind two barriers -> table
// words -> source values): PMC showed its waves WAITING 76 % of their 4.7 us life with the VALU 58 % busy.  Here a wave reads the
// flags of its whole tile at once, passes the sources through, compacts the tile's targets into a wave-private LDS list (ballot +
// popcount, no workgroup barrier) and then searches 64 targets per pass -- full waves whatever the failure rate, one wave start per
// ROWS rows, and the passes of the resident waves overlap each other's look-ups.
#ifndef HK_FILL_TILE_WAVES
#define HK_FILL_TILE_WAVES 8
#endif
template <int ROWS>
__global__ void __launch_bounds__(256, (ROWS <= 32 ? HK_FILL_TILE_WAVES : 5))  // (64-row tiles: the target lists' 32 KB of LDS per workgroup allow five)
inpaint_fill_tile_kernel(const float* __restrict__ offset, const unsigned char* __restrict__ flag,
                                                                   long long stride, int height, int width, int max_dist_arg,
                                                                   const unsigned* __restrict__ tb,
                                                                   const unsigned* __restrict__ tie,
                                                                   const double* __restrict__ wtab, float* __restrict__ filled) {
    static_assert(ROWS <= 64, "a list entry packs the tile row beside the lane; the row masks are 64-bit");
    __shared__ unsigned short lst[256 / WAVE][ROWS * WAVE];
    const int tid = threadIdx.x, lane = tid & (WAVE - 1), wv = tid / WAVE;
    const int x0 = blockIdx.x * blockDim.x + wv * WAVE, x_own = x0 + lane;
    const unsigned long long lt = (1ull << lane) - 1ull;
    // the search distance as a compile-time constant (the launcher passes FILL_MAX_DIST and nothing else): the first five steps become
    // straight-line code, the "nothing found yet" and acceptance bounds immediates
    const int max_dist = FILL_MAX_DIST;
    (void)max_dist_arg;
    // wave-uniform: the widest reach of a search stays inside the row -- the group that starts at step max_dist requests the one
    // after it (steps max_dist + 4 .. + 7) ahead, and a group's wide load covers 4 entries: 7 + 4 columns beyond max_dist
    const bool interior = x0 - max_dist - 12 >= 0 && x0 + WAVE - 1 + max_dist + 12 < width;
    const int n_tiles = (height + ROWS - 1) / ROWS;
    for (int tile = blockIdx.y; tile < n_tiles; tile += gridDim.y) {
        const int y0 = tile * ROWS;
        // flags of the whole tile: ROWS independent byte loads per lane, kept as two bit masks (bit r = row y0 + r)
        unsigned long long src_rows = 0ull, tgt_rows = 0ull;
#pragma unroll 8
        for (int r = 0; r < ROWS; ++r) {
            const bool in = x_own < width && y0 + r < height;
            const unsigned char f = in ? flag[(long long)(y0 + r) * stride + x_own] : (unsigned char)2;  // 2: outside the raster
            src_rows |= (unsigned long long)(f == 1) << r;
            tgt_rows |= (unsigned long long)(f == 0) << r;
        }
        int n = 0;  // wave-uniform: targets of the tile so far
#pragma unroll 4
        for (int r = 0; r < ROWS; ++r) {
            const long long i = (long long)(y0 + r) * stride + x_own;
            if ((src_rows >> r) & 1ull) filled[i] = offset[i];  // filled pixels never act as sources: sources pass through
            const bool target = (tgt_rows >> r) & 1ull;
            const unsigned long long bal = __ballot(target);
            if (target) lst[wv][n + (int)__popcll(bal & lt)] = (unsigned short)((r << 6) | lane);
            n += (int)__popcll(bal);
        }
        __syncthreads();  // (the list is the wave's own; the barrier only orders its LDS writes before the reads below)
        // passes of 64 targets (an explicit request of pass p + 1's first look-ups while pass p searches was measured slower:
        // 21.7 against 21.2 ms per step, profiles/r03_fill_tile.txt)
        for (int p = 0; p < n; p += WAVE) {
            if (p + lane < n) {
                const unsigned e = lst[wv][p + lane];
                const int x = x0 + (int)(e & 63u), y = y0 + (int)(e >> 6);
                const long long row = (long long)y * stride;
                filled[row + x] = interior ? fill_one<true>(x, y, row, offset, stride, width, max_dist, tb, tie, wtab)
                                           : fill_one<false>(x, y, row, offset, stride, width, max_dist, tb, tie, wtab);
            }
        }
        __syncthreads();  // the next tile re-uses the list
    }
}

// workspace: the distance table (2 of its 4 bytes per pixel are used) + source flags (1 byte per pixel) + the tie bitmap + the weight table
// + the column bit words (one 64-bit word per column and 64 rows; a plane of a few rows has more of those than spare table bytes)
static size_t bit_words(int height, long long stride) { return (size_t)((height + WORD_ROWS - 1) / WORD_ROWS) * (size_t)stride; }
size_t inpaint_workspace_bytes(int height, long long stride) {
    return (size_t)height * stride * 5 + 1024 + TIE_N / 8 + 256 + WTAB_N * 8 + 256 + bit_words(height, stride) * 8 + 256;
}

// the workspace's source-flag plane: the fit kernel can write it itself (FitArgs::flag), then gain / r2 are not needed here
unsigned char* inpaint_flag_plane(void* workspace, int height, long long stride) {
    return reinterpret_cast<unsigned char*>(static_cast<unsigned short*>(workspace) + 2 * (size_t)height * stride);
}

hipError_t launch_inpaint_offsets(const float* offset, const float* gain, const float* r2, float thresh, long long stride,
                                  int height, int width, void* workspace, float* filled, hipStream_t stream,
                                  const unsigned char* flag_ready, unsigned long long n_targets) {
    const size_t plane = (size_t)height * stride;
    unsigned* tb = static_cast<unsigned*>(workspace);  // (down^2 << 16) | up^2 row distances, 4 bytes per pixel
    unsigned char* ws_flag = inpaint_flag_plane(workspace, height, stride);
    const unsigned char* flag = flag_ready ? flag_ready : ws_flag;
    unsigned* tie = reinterpret_cast<unsigned*>(ws_flag + (plane + 255) / 256 * 256);
    double* wtab = reinterpret_cast<double*>(tie + (TIE_N / 32 + 64) / 64 * 64);
    hipLaunchKernelGGL(tie_kernel, dim3((TIE_N / 32 + 255) / 256), dim3(256), 0, stream, tie, wtab);
    const int max_dist = FILL_MAX_DIST;
    static_assert(100 + 1 < (int)NONE_B, "the table's distance bytes");
    if (!flag_ready)  // else: the flag plane was written by the fit kernel (FitArgs::flag)
        hipLaunchKernelGGL(inpaint_flag_kernel, dim3((width + 255) / 256, height < 1024 ? height : 1024), dim3(256), 0, stream,
                           gain, r2, thresh, stride, height, width, ws_flag);
    // column bit words behind the weight table (256-byte aligned)
    unsigned long long* bits = reinterpret_cast<unsigned long long*>(
        (reinterpret_cast<uintptr_t>(wtab + WTAB_N) + 255) / 256 * 256);
    const dim3 gbits((width + 1023) / 1024, (height + WORD_ROWS - 1) / WORD_ROWS);  // four columns per thread
    hipLaunchKernelGGL(inpaint_bits_kernel, gbits, dim3(256), 0, stream, flag, stride, height, width, bits);
    const dim3 gtable((width + 511) / 512, (height + WORD_ROWS - 1) / WORD_ROWS);  // two columns per thread
    hipLaunchKernelGGL(inpaint_table_kernel, gtable, dim3(256), 0, stream, bits, stride, height, width, max_dist, tb);
    const dim3 gfill((width + 255) / 256, height < 65535 ? height : 65535);
    // The TILED search (64 columns x 32 rows per wave; 16 rows 21.4 against 21.2 ms per step, 64 rows 27.1 when it was introduced,
    // profiles/r03_fill_tile.txt) serves every failure rate since its candidate test is branch-free (at 94 % failures 36.7 against
    // 37.7 ms for the one-thread-per-pixel form).  HK_FILL_TILE=0 selects the per-row forms of rounds 1-2 (A/B; n_targets -- the
    // number of pixels to fill as the caller knows it, 0 = unknown -- picks between them), 4..64 another tile height.
    static const int tile_env = [] { const char* e = getenv("HK_FILL_TILE"); return e ? atoi(e) : -1; }();
    const bool moderate = n_targets == 0 || (double)n_targets < 0.6 * (double)height * (double)width;
    const int tile_rows = tile_env >= 0 ? tile_env : 32;
    if (tile_rows > 0) {
        const int rows = tile_rows >= 64 ? 64 : (tile_rows >= 32 ? 32 : (tile_rows >= 16 ? 16 : (tile_rows >= 8 ? 8 : 4)));
        const int n_tiles = (height + rows - 1) / rows;
        const dim3 gt((width + 255) / 256, n_tiles < 65535 ? n_tiles : 65535);
        if (rows == 64)
            hipLaunchKernelGGL(inpaint_fill_tile_kernel<64>, gt, dim3(256), 0, stream, offset, flag, stride, height, width, max_dist, tb, tie, wtab, filled);
        else if (rows == 32)
            hipLaunchKernelGGL(inpaint_fill_tile_kernel<32>, gt, dim3(256), 0, stream, offset, flag, stride, height, width, max_dist, tb, tie, wtab, filled);
        else if (rows == 16)
            hipLaunchKernelGGL(inpaint_fill_tile_kernel<16>, gt, dim3(256), 0, stream, offset, flag, stride, height, width, max_dist, tb, tie, wtab, filled);
        else if (rows == 8)
            hipLaunchKernelGGL(inpaint_fill_tile_kernel<8>, gt, dim3(256), 0, stream, offset, flag, stride, height, width, max_dist, tb, tie, wtab, filled);
        else
            hipLaunchKernelGGL(inpaint_fill_tile_kernel<4>, gt, dim3(256), 0, stream, offset, flag, stride, height, width, max_dist, tb, tie, wtab, filled);
        return hipGetLastError();
    }
    if (moderate)
        hipLaunchKernelGGL(inpaint_fill_kernel<true>, gfill, dim3(256), 0, stream, offset, flag, stride, height, width, max_dist,
                           tb, tie, wtab, filled);
    else
        hipLaunchKernelGGL(inpaint_fill_kernel<false>, gfill, dim3(256), 0, stream, offset, flag, stride, height, width, max_dist,
                           tb, tie, wtab, filled);
    return hipGetLastError();
}

}  // namespace hk
